// dev_util.hpp — small device/host helpers shared by the gfx950 kernels (wave64 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>

namespace edsx {

struct DeviceError : std::runtime_error { using std::runtime_error::runtime_error; };
struct OutOfDeviceMemory : DeviceError { using DeviceError::DeviceError; };     // hipMalloc said so: the caller may retry with less

#define EDSX_HIP(call)                                                                        \
    do {                                                                                      \
        hipError_t e__ = (call);                                                              \
        if (e__ != hipSuccess)                                                                \
            throw ::edsx::DeviceError(std::string(#call) + ": " + hipGetErrorString(e__));    \
    } while (0)

typedef unsigned long long u64;
typedef unsigned int u32;

// ---- large device-to-host copies ----------------------------------------------------------------------
// hipMemcpy into pageable memory is copied out of the runtime's staging buffer by ONE host thread (≈18 GB/s into
// huge-page-backed memory on this stack), well under what the link gives a pinned destination.  Outputs of this
// engine are GB-sized text, so large downloads go through two pinned staging chunks: the DMA of chunk i+1 overlaps
// the copy of chunk i into the caller's buffer by several host threads.  Synchronous: returns when dst is complete.
class PinnedDownload {
public:
    static void copy(void* dst, const void* dev_src, size_t n, hipStream_t st)
    {
        if (n < MIN_BYTES) {
            if (n) EDSX_HIP(hipMemcpyAsync(dst, dev_src, n, hipMemcpyDeviceToHost, st));
            EDSX_HIP(hipStreamSynchronize(st));
            return;
        }
        // one set of staging buffers and events per device (events belong to the device that is current when they
        // are created; contexts on different GPUs of one process download side by side)
        int dev = 0;
        EDSX_HIP(hipGetDevice(&dev));
        static Stage stages[MAX_DEVICES];
        if (dev < 0 || dev >= MAX_DEVICES) {
            EDSX_HIP(hipMemcpyAsync(dst, dev_src, n, hipMemcpyDeviceToHost, st));
            EDSX_HIP(hipStreamSynchronize(st));
            return;
        }
        Stage& sg = stages[dev];
        std::lock_guard<std::mutex> lock(sg.mu);
        sg.ensure();
        uint8_t* out = static_cast<uint8_t*>(dst);
        const uint8_t* src = static_cast<const uint8_t*>(dev_src);
        const size_t nchunks = (n + CHUNK - 1) / CHUNK;
        auto issue = [&](size_t i) {
            const size_t off = i * CHUNK, len = std::min(CHUNK, n - off);
            EDSX_HIP(hipMemcpyAsync(sg.pin[i & 1], src + off, len, hipMemcpyDeviceToHost, st));
            EDSX_HIP(hipEventRecord(sg.ev[i & 1], st));
        };
        issue(0);
        for (size_t i = 0; i < nchunks; i++) {
            if (i + 1 < nchunks) issue(i + 1);
            EDSX_HIP(hipEventSynchronize(sg.ev[i & 1]));
            const size_t off = i * CHUNK, len = std::min(CHUNK, n - off);
            const uint8_t* from = sg.pin[i & 1];
            std::thread th[THREADS - 1];
            const size_t part = ((len + THREADS - 1) / THREADS + 63) & ~(size_t)63;     // THREADS * part >= len
            for (unsigned t = 1; t < THREADS; t++) {
                const size_t lo = std::min(len, part * t), hi = std::min(len, part * (t + 1));
                th[t - 1] = std::thread([=] { if (hi > lo) std::memcpy(out + off + lo, from + lo, hi - lo); });
            }
            std::memcpy(out + off, from, std::min(len, part));
            for (auto& x : th) x.join();
        }
    }

private:
    static constexpr size_t MIN_BYTES = (size_t)16 << 20, CHUNK = (size_t)32 << 20;
    static constexpr unsigned THREADS = 8;
    static constexpr int MAX_DEVICES = 16;
    struct Stage {
        std::mutex mu;
        uint8_t* pin[2] = {nullptr, nullptr};
        hipEvent_t ev[2];
        bool ready = false;
        void ensure()
        {
            if (ready) return;
            for (int k = 0; k < 2; k++) {
                EDSX_HIP(hipHostMalloc(reinterpret_cast<void**>(&pin[k]), CHUNK, hipHostMallocPortable));
                EDSX_HIP(hipEventCreateWithFlags(&ev[k], hipEventDisableTiming));
            }
            ready = true;
        }
    };
};

// The device tokenisers keep a few u64 per input byte (flags and their scans): fine for the GB-sized texts they are
// meant for, but a text whose scratch would not fit comfortably is left to the host tokenisers instead of failing.
inline bool device_scratch_fits(size_t bytes)
{
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return false;
    return bytes <= free_b / 2;
}

// ---- wave64 primitives -------------------------------------------------------------------
__device__ __forceinline__ u32 lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
// number of set bits of `mask` strictly below this lane
__device__ __forceinline__ u32 mbcnt(u64 mask)
{
    return __builtin_amdgcn_mbcnt_hi((u32)(mask >> 32), __builtin_amdgcn_mbcnt_lo((u32)mask, 0u));
}
__device__ __forceinline__ u64 ballot64(bool p) { return __ballot(p); }

__device__ __forceinline__ u64 mix64(u64 x)
{   // splitmix64 finaliser
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__device__ __forceinline__ u64 hash3(u64 seed, u64 a, u64 b)
{
    return mix64(seed ^ mix64(a ^ mix64(b + 0x632BE59BD9B4E019ull)));
}

// 16 bytes from an arbitrarily aligned address (gfx950 global loads tolerate misalignment;
// hipcc emits one global_load_dwordx4 for this)
struct __attribute__((packed, aligned(1))) U128u { uint32_t x, y, z, w; };
__device__ __forceinline__ uint4 load16u(const uint8_t* p)
{
    U128u v;
    __builtin_memcpy(&v, p, 16);
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void store16u(uint8_t* p, const uint4& v)   // 16 bytes to an arbitrarily aligned address
{
    U128u t{v.x, v.y, v.z, v.w};
    __builtin_memcpy(p, &t, 16);
}
// partial load of n (<16) bytes, zero padded
__device__ __forceinline__ uint4 load_partial(const uint8_t* p, int n)
{
    u64 lo = 0, hi = 0;
    for (int i = 0; i < n; i++) {
        u64 b = p[i];
        if (i < 8) lo |= b << (8 * i);
        else hi |= b << (8 * (i - 8));
    }
    return make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
}

// per-byte "a != b" mask of a dword pair as 4 bits
__device__ __forceinline__ u32 ne_bytes4(uint32_t a, uint32_t b)
{
    uint32_t x = a ^ b;
    // byte != 0  <=>  ((x & 0x7f7f7f7f) + 0x7f7f7f7f | x) & 0x80808080
    uint32_t t = (((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u;
    // gather bits 7,15,23,31 -> 0..3
    return ((t >> 7) & 1u) | ((t >> 14) & 2u) | ((t >> 21) & 4u) | ((t >> 28) & 8u);
}
__device__ __forceinline__ u32 eq_byte4(uint32_t a, uint32_t cccc) { return 0xFu ^ ne_bytes4(a, cccc); }

__device__ __forceinline__ u32 byte_of(const uint4& v, int i)
{
    uint32_t w = (i < 8) ? ((i < 4) ? v.x : v.y) : ((i < 12) ? v.z : v.w);
    return (w >> ((i & 3) * 8)) & 0xffu;
}

// decimal digits of v (v >= 1)
__device__ __forceinline__ u32 ndigits(u32 v)
{
    return 1u + (v >= 10u) + (v >= 100u) + (v >= 1000u) + (v >= 10000u) + (v >= 100000u) +
           (v >= 1000000u) + (v >= 10000000u) + (v >= 100000000u) + (v >= 1000000000u);
}

// ---- device-wide scans on u64 arrays, element count read from device memory -----------------
// OP 0: exclusive sum (out[i] = sum(in[0..i)), *total = sum of all)
// OP 1: inclusive max (out[i] = max(in[0..i]),  *total = max of all)
// in/out may alias.  Three grid-stride kernels; tmp holds one u64 per SCAN_TILE elements.
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;   // 2048

template <int OP> __device__ __forceinline__ u64 scan_op(u64 a, u64 b) { return OP == 0 ? a + b : (a > b ? a : b); }

template <int OP>
__device__ __forceinline__ u64 block_reduce(u64 v, u64* sh /*[SCAN_THREADS/64]*/)
{
    for (int o = 32; o > 0; o >>= 1) v = scan_op<OP>(v, __shfl_down(v, o, 64));
    int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    u64 t = 0;
    for (int i = 0; i < (int)(blockDim.x >> 6); i++) t = scan_op<OP>(t, sh[i]);
    __syncthreads();
    return t;
}

template <int OP>
static __global__ void k_scan_tile_sums(const u64* __restrict__ in, const u64* __restrict__ n_ptr,
                                        u64* __restrict__ bsum)
{
    __shared__ u64 sh[SCAN_THREADS / 64];
    const u64 n = *n_ptr;
    const u64 ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    for (u64 t = blockIdx.x; t < ntiles; t += gridDim.x) {
        u64 base = t * SCAN_TILE;
        u64 s = 0;
        for (int i = 0; i < SCAN_ITEMS; i++) {
            u64 idx = base + (u64)i * SCAN_THREADS + threadIdx.x;
            if (idx < n) s = scan_op<OP>(s, in[idx]);
        }
        s = block_reduce<OP>(s, sh);
        if (threadIdx.x == 0) bsum[t] = s;
    }
}

// single workgroup: exclusive scan of bsum[0..ntiles) in place, total -> *total
template <int OP>
static __global__ void k_scan_spine(u64* __restrict__ bsum, const u64* __restrict__ n_ptr,
                                    u64* __restrict__ total)
{
    __shared__ u64 sh[1024];
    __shared__ u64 carry_sh;
    const u64 n = *n_ptr;
    const u64 ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    if (threadIdx.x == 0) carry_sh = 0;
    __syncthreads();
    for (u64 base = 0; base < ntiles; base += blockDim.x) {
        u64 idx = base + threadIdx.x;
        u64 v = idx < ntiles ? bsum[idx] : 0;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < (int)blockDim.x; o <<= 1) {       // Hillis-Steele inclusive
            u64 a = threadIdx.x >= (u32)o ? sh[threadIdx.x - o] : 0;
            __syncthreads();
            sh[threadIdx.x] = scan_op<OP>(sh[threadIdx.x], a);
            __syncthreads();
        }
        u64 incl = sh[threadIdx.x];
        u64 excl = threadIdx.x ? sh[threadIdx.x - 1] : 0;
        u64 carry = carry_sh;
        if (idx < ntiles) bsum[idx] = scan_op<OP>(carry, excl);
        __syncthreads();
        if (threadIdx.x == blockDim.x - 1) carry_sh = scan_op<OP>(carry, incl);
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry_sh;
}

template <int OP>
static __global__ void k_scan_apply(const u64* __restrict__ in, u64* __restrict__ out,
                                    const u64* __restrict__ n_ptr, const u64* __restrict__ bsum)
{
    __shared__ u64 wsum[SCAN_THREADS / 64];
    const u64 n = *n_ptr;
    const u64 ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (u64 t = blockIdx.x; t < ntiles; t += gridDim.x) {
        // blocked arrangement: thread owns SCAN_ITEMS consecutive items
        u64 base = t * SCAN_TILE + (u64)threadIdx.x * SCAN_ITEMS;
        u64 v[SCAN_ITEMS];
        u64 s = 0;
        for (int i = 0; i < SCAN_ITEMS; i++) {
            v[i] = (base + i < n) ? in[base + i] : 0;
            s = scan_op<OP>(s, v[i]);
        }
        u64 incl = s;                                           // wave inclusive scan of s
        for (int o = 1; o < 64; o <<= 1) {
            u64 a = __shfl_up(incl, o, 64);
            if (lane >= o) incl = scan_op<OP>(incl, a);
        }
        u64 excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = 0;
        if (lane == 63) wsum[w] = incl;
        __syncthreads();
        u64 woff = 0;
        for (int i = 0; i < w; i++) woff = scan_op<OP>(woff, wsum[i]);
        u64 run = scan_op<OP>(scan_op<OP>(bsum[t], woff), excl);   // everything before this thread's items
        for (int i = 0; i < SCAN_ITEMS; i++) {
            if (OP == 0) { if (base + i < n) out[base + i] = run; run += v[i]; }
            else { run = scan_op<OP>(run, v[i]); if (base + i < n) out[base + i] = run; }
        }
        __syncthreads();
    }
}

template <int OP>
inline void device_scan_u64(const u64* in, u64* out, const u64* d_n, u64* d_total, u64* tmp, hipStream_t st)
{
    hipLaunchKernelGGL((k_scan_tile_sums<OP>), dim3(1024), dim3(SCAN_THREADS), 0, st, in, d_n, tmp);
    hipLaunchKernelGGL((k_scan_spine<OP>), dim3(1), dim3(1024), 0, st, tmp, d_n, d_total);
    hipLaunchKernelGGL((k_scan_apply<OP>), dim3(1024), dim3(SCAN_THREADS), 0, st, in, out, d_n, tmp);
}
inline void exclusive_scan_u64(const u64* in, u64* out, const u64* d_n, u64* d_total, u64* tmp, hipStream_t st)
{
    device_scan_u64<0>(in, out, d_n, d_total, tmp, st);
}

// ---- N exclusive sums over arrays of the same length in ONE pass of the three kernels (the MSA planner scans two or
// three size / flag arrays per step: a third of the launches and of the tail latencies).  tmp: N * (n / SCAN_TILE + 2) u64.
template <int N> struct ScanSet { const u64* in[N]; u64* out[N]; u64* total[N]; };

template <int N>
static __global__ void k_mscan_tile_sums(ScanSet<N> a, const u64* __restrict__ n_ptr, u64* __restrict__ bsum)
{
    __shared__ u64 sh[SCAN_THREADS / 64];
    const u64 n = *n_ptr;
    const u64 ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    for (u64 t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const u64 base = t * SCAN_TILE;
        u64 s[N];
#pragma unroll
        for (int k = 0; k < N; k++) s[k] = 0;
        for (int i = 0; i < SCAN_ITEMS; i++) {
            const u64 idx = base + (u64)i * SCAN_THREADS + threadIdx.x;
            if (idx < n) {
#pragma unroll
                for (int k = 0; k < N; k++) s[k] += a.in[k][idx];
            }
        }
#pragma unroll
        for (int k = 0; k < N; k++) {
            const u64 r = block_reduce<0>(s[k], sh);
            if (threadIdx.x == 0) bsum[(u64)k * (ntiles + 1) + t] = r;
        }
    }
}
template <int N>
static __global__ void k_mscan_spine(ScanSet<N> a, u64* __restrict__ bsum, const u64* __restrict__ n_ptr)
{
    // N waves, one per array: a wave walks its tile sums 64 at a time (DPP-free shuffle scan)
    const u64 n = *n_ptr;
    const u64 ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    const int k = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (k >= N) return;
    u64* b = bsum + (u64)k * (ntiles + 1);
    u64 carry = 0;
    for (u64 base = 0; base < ntiles; base += 64) {
        const u64 idx = base + lane;
        const u64 v = idx < ntiles ? b[idx] : 0;
        u64 incl = v;
        for (int o = 1; o < 64; o <<= 1) { const u64 x = __shfl_up(incl, o, 64); if (lane >= o) incl += x; }
        if (idx < ntiles) b[idx] = carry + incl - v;
        carry += __shfl(incl, 63, 64);
    }
    if (lane == 0) *a.total[k] = carry;
}
template <int N>
static __global__ void k_mscan_apply(ScanSet<N> a, const u64* __restrict__ n_ptr, const u64* __restrict__ bsum)
{
    __shared__ u64 wsum[N][SCAN_THREADS / 64];
    const u64 n = *n_ptr;
    const u64 ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (u64 t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const u64 base = t * SCAN_TILE + (u64)threadIdx.x * SCAN_ITEMS;
        u64 v[N][SCAN_ITEMS], excl[N];
#pragma unroll
        for (int k = 0; k < N; k++) {
            u64 s = 0;
            for (int i = 0; i < SCAN_ITEMS; i++) { v[k][i] = (base + i < n) ? a.in[k][base + i] : 0; s += v[k][i]; }
            u64 incl = s;
            for (int o = 1; o < 64; o <<= 1) { const u64 x = __shfl_up(incl, o, 64); if (lane >= o) incl += x; }
            excl[k] = incl - s;
            if (lane == 63) wsum[k][w] = incl;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < N; k++) {
            u64 woff = 0;
            for (int i = 0; i < w; i++) woff += wsum[k][i];
            u64 run = bsum[(u64)k * (ntiles + 1) + t] + woff + excl[k];
            for (int i = 0; i < SCAN_ITEMS; i++) { if (base + i < n) a.out[k][base + i] = run; run += v[k][i]; }
        }
        __syncthreads();
    }
}
template <int N>
inline void exclusive_scan_multi(const ScanSet<N>& a, const u64* d_n, u64* tmp, hipStream_t st)
{
    hipLaunchKernelGGL((k_mscan_tile_sums<N>), dim3(1024), dim3(SCAN_THREADS), 0, st, a, d_n, tmp);
    hipLaunchKernelGGL((k_mscan_spine<N>), dim3(1), dim3(64 * N), 0, st, a, tmp, d_n);
    hipLaunchKernelGGL((k_mscan_apply<N>), dim3(1024), dim3(SCAN_THREADS), 0, st, a, d_n, tmp);
}
inline void inclusive_max_scan_u64(const u64* in, u64* out, const u64* d_n, u64* d_total, u64* tmp, hipStream_t st)
{
    device_scan_u64<1>(in, out, d_n, d_total, tmp, st);
}

} // namespace edsx
