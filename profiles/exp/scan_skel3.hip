// scan_skel3.hip — round-3 experiment 3: can a low-register prefetcher (LDS-DMA into a dummy LDS page, 1 KB of one row per
// wave-instruction = DRAM-friendly) feed the column scan from the Infinity Cache / L2 instead of HBM?
//   P1  the prefetcher alone over the whole image (HBM -> nowhere rate of 1-KB row pieces at 1 or 2 small workgroups per CU)
//   P2  skeleton over a 131 MB column range: cold, and right after the prefetcher has touched that range
//   P3  prefetcher and skeleton side by side on two streams, the prefetcher throttled to stay AHEAD tiles in front of the
//       skeleton's progress counters
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o scan_skel3 scan_skel3.hip && ./scan_skel3 [S] [L] [reps]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef unsigned long long u64;
typedef unsigned int u32;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ u64 mix64(u64 x) { x += 0x9e3779b97f4a7c15ull; x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull; x = (x ^ (x >> 27)) * 0x94d049bb133111ebull; return x ^ (x >> 31); }
__device__ __forceinline__ uint8_t cell(u64 r, u64 c)
{
    const u64 h = mix64(c);
    uint8_t ref = "ACGT"[h & 3];
    if ((h >> 8) % 20 == 0 && r) { const u64 g = mix64(c * 1315423911ull + r); if (g & 1) ref = "ACGT"[(h + 1 + (g >> 1) % 3) & 3]; }
    return ref;
}
__global__ void k_fill(uint8_t* f, u64 S, u64 L, u64 stride, u64 off0)
{
    const u64 n = S * L;
    for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i < n / 4; i += (u64)gridDim.x * blockDim.x) {
        const u64 r = (i * 4) / L, c = (i * 4) % L;
        uint8_t* p = f + off0 + r * stride + c;
        p[0] = cell(r, c); p[1] = cell(r, c + 1); p[2] = cell(r, c + 2); p[3] = cell(r, c + 3);
    }
}
__global__ void k_ref_mask(const uint8_t* f, u64 S, u64 L, u64 stride, u64 off0, u64* V)
{
    const u64 c = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    bool var = false;
    if (c < L) { const uint8_t r0 = f[off0 + c]; for (u64 r = 1; r < S; r++) var |= f[off0 + r * stride + c] != r0; }
    const u64 b = __ballot(var);
    if ((threadIdx.x & 63) == 0 && c < L) V[c >> 6] = b;
}
__device__ __forceinline__ uint4 load16u(const uint8_t* p) { uint4 v; __builtin_memcpy(&v, p, 16); return v; }
__device__ __forceinline__ u32 nz4(u32 x) { u32 h = (((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u; return ((h >> 7) | (h >> 14) | (h >> 21) | (h >> 28)) & 0xfu; }
__device__ __forceinline__ u32 nz16(u32 x, u32 y, u32 z, u32 w) { return nz4(x) | (nz4(y) << 4) | (nz4(z) << 8) | (nz4(w) << 12); }

// tile of workgroup b: XCD b % 8 owns a contiguous range of the tiles [t0, t0 + nt)
__device__ __forceinline__ u64 tile_of(u64 b, u64 t0, u64 nt, u64* kout, u64* xout)
{
    const u64 per = nt / 8, rem = nt % 8, x = b % 8, k = b / 8;
    *kout = k; *xout = x;
    return t0 + x * per + (x < rem ? x : rem) + k;
}

// ---- skeleton (one workgroup per tile; dynamic LDS limits the workgroups per CU); done[x] counts finished tiles of XCD range x
__global__ void __launch_bounds__(512, 4) k_skel(const uint8_t* __restrict__ f, const u64* __restrict__ row_start, u64 t0, u64 nt,
                                                 u64* __restrict__ V, u64* done)
{
    extern __shared__ uint8_t occupancy_pad[];
    __shared__ u32 D[8];
    const u32 tid = threadIdx.x, j = tid & 7, sub = tid >> 3;
    if (tid < 8) D[tid] = 0;
    if (nt == 0) occupancy_pad[tid] = 0;
    u64 k, x;
    const u64 tile = tile_of(blockIdx.x, t0, nt, &k, &x);
    const u64 q = tile * 128 + j * 16;
    const uint4 ref = load16u(f + row_start[0] + q);
    const ulonglong2* rp = reinterpret_cast<const ulonglong2*>(row_start + sub * 16u);
    ulonglong2 rv[8];
#pragma unroll
    for (int i = 0; i < 8; i++) rv[i] = rp[i];
    uint4 d[16];
#pragma unroll
    for (int i = 0; i < 8; i++) { d[2 * i] = load16u(f + rv[i].x + q); d[2 * i + 1] = load16u(f + rv[i].y + q); }
    uint4 acc = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int it = 0; it < 16; it++) { acc.x |= d[it].x ^ ref.x; acc.y |= d[it].y ^ ref.y; acc.z |= d[it].z ^ ref.z; acc.w |= d[it].w ^ ref.w; }
    u32 diff = nz16(acc.x, acc.y, acc.z, acc.w);
    for (u32 o = 8; o < 64u; o <<= 1) diff |= (u32)__shfl_xor((int)diff, (int)o, 64);
    __syncthreads();
    if ((tid & 63u) < 8 && diff) atomicOr(&D[j], diff);
    __syncthreads();
    if (tid < 2) V[tile * 2 + tid] = (u64)D[4 * tid] | ((u64)D[4 * tid + 1] << 16) | ((u64)D[4 * tid + 2] << 32) | ((u64)D[4 * tid + 3] << 48);
    if (done && tid == 0) atomicAdd(&done[x * 16], 1ull);                 // (one counter per 128-byte line)
}

// ---- prefetcher: wave-instruction = 1 KB of ONE row (64 lanes x 16 B) -> dummy LDS page.  Items in unit-major order
// (unit = 8 tiles = 1 KB of every row) per XCD range, so it advances along the columns like the skeleton does.
template <bool DMA>
__global__ void __launch_bounds__(256) k_prefetch(const uint8_t* __restrict__ f, const u64* __restrict__ row_start, u32 S, u64 t0, u64 nt,
                                                  const u64* done, u64 ahead_tiles, u64* sink)
{
    __shared__ __attribute__((aligned(1024))) uint8_t dummy[4][1024];
    const u32 lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const u64 wave = blockIdx.x * 4ull + wv, nwaves = gridDim.x * 4ull;
    const u64 x = wave % 8, wr = wave / 8, nwr = nwaves / 8;                // this wave's XCD range, its index among the range's waves
    const u64 per = nt / 8, rem = nt % 8;
    const u64 tb = t0 + x * per + (x < rem ? x : rem), tn = per + (x < rem ? 1 : 0);
    const u64 nunits = (tn + 7) / 8;
    u32 acc = 0;
    for (u64 u = 0; u < nunits; u++) {
        if (done) {                                                       // stay at most ahead_tiles in front of the finished tiles
            u32 spins = 0;                                                // bounded: if the skeleton never runs beside us, go on
            while (__hip_atomic_load(&done[x * 16], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + ahead_tiles < 8 * u && spins < 200000u) {
                __builtin_amdgcn_s_sleep(32); spins++;
            }
        }
        const u64 colb = (tb + 8 * u) * 128 + (u64)lane * 16;
        for (u64 r = wr; r < S; r += nwr) {
            const uint8_t* g = f + row_start[r] + colb;
            if constexpr (DMA)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                 (__attribute__((address_space(3))) void*)dummy[wv], 16, 0, 0);
            else { const uint4 v = load16u(g); acc |= v.x ^ v.y ^ v.z ^ v.w; }
        }
    }
    if (acc == 0x12345u) sink[0] = acc;
}

static double time_ms(hipEvent_t a, hipEvent_t b) { float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms; }

int main(int argc, char** argv)
{
    const u64 S = argc > 1 ? strtoull(argv[1], 0, 10) : 1000;
    const u64 L = argc > 2 ? strtoull(argv[2], 0, 10) : 20000000;       // multiple of 1024
    const int reps = argc > 3 ? atoi(argv[3]) : 3;
    const u64 stride = L + 7, off0 = 5;
    const u64 nbytes = off0 + S * stride + (1 << 20);
    uint8_t* f; CK(hipMalloc(&f, nbytes));
    CK(hipMemset(f, 'A', nbytes));
    k_fill<<<4096, 256>>>(f, S, L, stride, off0);
    const u64 SPmax = 1024 + 16;
    std::vector<u64> rs(SPmax);
    for (u64 r = 0; r < SPmax; r++) rs[r] = off0 + (r < S ? r : S - 1) * stride;
    u64* d_rs; CK(hipMalloc(&d_rs, SPmax * 8)); CK(hipMemcpy(d_rs, rs.data(), SPmax * 8, hipMemcpyHostToDevice));
    const u64 nwords = L / 64;
    u64 *Vref, *V, *done, *sink; CK(hipMalloc(&Vref, nwords * 8)); CK(hipMalloc(&V, nwords * 8)); CK(hipMalloc(&done, 8 * 16 * 8)); CK(hipMalloc(&sink, 64));
    k_ref_mask<<<(unsigned)((L + 255) / 256), 256>>>(f, S, L, stride, off0, Vref);
    CK(hipDeviceSynchronize());
    std::vector<u64> href(nwords), hv(nwords);
    CK(hipMemcpy(href.data(), Vref, nwords * 8, hipMemcpyDeviceToHost));
    printf("S=%llu L=%llu bytes=%.2f GB\n", S, L, S * L / 1e9);
    hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    hipStream_t sP, sS; CK(hipStreamCreateWithFlags(&sP, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sS, hipStreamNonBlocking));
    const size_t OCC2 = 64 * 1024;
    CK(hipFuncSetAttribute((const void*)k_skel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)OCC2));
    const u64 ntiles = L / 128;
    auto report = [&](const char* name, double ms, double bytes, u64 t0, u64 nt, bool mask) {
        u64 bad = 0;
        if (mask) { CK(hipMemcpy(hv.data(), V, nwords * 8, hipMemcpyDeviceToHost)); for (u64 i = t0 * 2; i < (t0 + nt) * 2; i++) bad += hv[i] != href[i]; }
        printf("%-66s %8.3f ms  %6.2f TB/s  %s\n", name, ms, bytes / ms / 1e9, !mask ? "" : bad ? "mask WRONG" : "mask ok");
        fflush(stdout);
        CK(hipMemset(V, 0, nwords * 8));
    };
    // ---- P1
    for (int g : {256, 512, 1024}) {
        for (int dma = 1; dma >= 0; dma--) {
            double best = 1e30;
            for (int i = 0; i < reps; i++) {
                CK(hipEventRecord(e0, sP));
                if (dma) k_prefetch<true><<<g, 256, 0, sP>>>(f, d_rs, (u32)S, 0, ntiles, nullptr, 0, sink);
                else k_prefetch<false><<<g, 256, 0, sP>>>(f, d_rs, (u32)S, 0, ntiles, nullptr, 0, sink);
                CK(hipEventRecord(e1, sP)); CK(hipEventSynchronize(e1)); best = std::min(best, time_ms(e0, e1));
            }
            char nm[128]; snprintf(nm, sizeof nm, "P1 prefetcher alone, %d workgroups of 4 waves, %s", g, dma ? "LDS-DMA" : "register loads");
            report(nm, best, (double)S * L, 0, 0, false);
        }
    }
    // ---- P2: 1024 tiles = 131 MB
    {
        const u64 nt = 1024;
        for (int mode = 0; mode < 3; mode++) {
            double best = 1e30;
            for (int i = 0; i < reps; i++) {
                const u64 t0 = 8192 * (u64)(1 + i + 4 * mode);                // a fresh range every time
                if (mode == 1) k_prefetch<true><<<1024, 256, 0, sS>>>(f, d_rs, (u32)S, t0, nt, nullptr, 0, sink);
                if (mode == 2) k_prefetch<false><<<1024, 256, 0, sS>>>(f, d_rs, (u32)S, t0, nt, nullptr, 0, sink);
                CK(hipEventRecord(e0, sS));
                k_skel<<<(unsigned)nt, 512, OCC2, sS>>>(f, d_rs, t0, nt, V, nullptr);
                CK(hipEventRecord(e1, sS)); CK(hipEventSynchronize(e1)); best = std::min(best, time_ms(e0, e1));
                if (i == reps - 1) report(mode == 0 ? "P2 skeleton over 131 MB, cold" : mode == 1 ? "P2 skeleton over 131 MB just touched by LDS-DMA" :
                                          "P2 skeleton over 131 MB just touched by register loads", best, (double)S * nt * 128, t0, nt, true);
            }
        }
    }
    // ---- P3
    {
        double best = 1e30;
        for (int i = 0; i < reps; i++) {
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, sS));
            k_skel<<<(unsigned)ntiles, 512, OCC2, sS>>>(f, d_rs, 0, ntiles, V, nullptr);
            CK(hipEventRecord(e1, sS)); CK(hipEventSynchronize(e1)); best = std::min(best, time_ms(e0, e1));
        }
        report("P3 control: skeleton alone, 2 per CU", best, (double)S * L, 0, ntiles, true);
        for (int dma = 1; dma >= 0; dma--)
            for (u64 ahead : {256ull, 1024ull, 4096ull})                  // tiles per XCD range: x 8 ranges x 128 KB
                for (int g : {256, 512}) {
                    best = 1e30;
                    for (int i = 0; i < reps; i++) {
                        CK(hipMemsetAsync(done, 0, 8 * 16 * 8, sS));
                        CK(hipDeviceSynchronize());
                        CK(hipEventRecord(e0, sS));
                        CK(hipStreamWaitEvent(sP, e0, 0));
                        if (dma) k_prefetch<true><<<g, 256, 0, sP>>>(f, d_rs, (u32)S, 0, ntiles, done, ahead, sink);
                        else k_prefetch<false><<<g, 256, 0, sP>>>(f, d_rs, (u32)S, 0, ntiles, done, ahead, sink);
                        k_skel<<<(unsigned)ntiles, 512, OCC2, sS>>>(f, d_rs, 0, ntiles, V, done);
                        CK(hipEventRecord(e2, sP));
                        CK(hipStreamWaitEvent(sS, e2, 0));
                        CK(hipEventRecord(e1, sS)); CK(hipEventSynchronize(e1)); best = std::min(best, time_ms(e0, e1));
                    }
                    char nm[160];
                    snprintf(nm, sizeof nm, "P3 skeleton + %s prefetcher (%d wg), %llu tiles (%.0f MB) ahead", dma ? "LDS-DMA" : "register", g, ahead, ahead * 8 * 0.128);
                    report(nm, best, (double)S * L, 0, ntiles, true);
                }
    }
    return 0;
}
