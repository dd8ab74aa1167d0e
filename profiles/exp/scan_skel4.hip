// scan_skel4.hip — round-3 experiment 4: what in the ACCESS PATTERN holds the register skeleton at 4.6-4.7 TB/s when a
// linear stream runs at 6.5?  Same loads + reduce (512 threads, thread = 16 x 16 B), different address patterns:
//   LINu   linear stream from a byte-unaligned base
//   A      1000 rows x 128-byte pieces, rows byte-unaligned (the real case)
//   Aa128  ... every row piece 128-byte aligned       Aa16  ... 16-byte aligned only
//   W<k>   wider pieces from fewer rows at equal bytes per workgroup: (1000/k) rows x (128 k) bytes (k = 2, 4, 8) - mask garbage, timing only
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o scan_skel4 scan_skel4.hip && ./scan_skel4
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned long long u64;
typedef unsigned int u32;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__device__ __forceinline__ uint4 load16u(const uint8_t* p) { uint4 v; __builtin_memcpy(&v, p, 16); return v; }

__global__ void k_fillw(u32* f, u64 nwords) { for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i < nwords; i += (u64)gridDim.x * blockDim.x) f[i] = (u32)(i * 2654435761u) & 0x03030303u; }

__global__ void __launch_bounds__(512, 4) k_linear(const uint8_t* __restrict__ f, u64* __restrict__ out)
{
    extern __shared__ uint8_t pad[];
    const uint8_t* base = f + (u64)blockIdx.x * 131072 + threadIdx.x * 16;
    uint4 d[16];
#pragma unroll
    for (int i = 0; i < 16; i++) d[i] = load16u(base + i * 8192);
    u32 a = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) a |= d[i].x ^ d[i].y ^ d[i].z ^ d[i].w;
    if (a == 0x12345678u) { out[blockIdx.x] = a; pad[threadIdx.x] = 1; }
}

// rows_log2 chunks per row piece = 8 << wl (piece = 128 << wl bytes); a thread holds 16 rows x 16 B; the workgroup covers
// (64 >> wl) row groups of 16 rows.  Tiles advance along the columns inside an XCD-owned range like the real kernel.
__global__ void __launch_bounds__(512, 4) k_skel(const uint8_t* __restrict__ f, const u64* __restrict__ row_start, u32 wl, u64 ntiles,
                                                 u64* __restrict__ V)
{
    extern __shared__ uint8_t pad[];
    __shared__ u32 D[64];
    const u32 tid = threadIdx.x, cpr = 8u << wl, j = tid & (cpr - 1), sub = tid >> (3 + wl);
    if (tid < 64) D[tid] = 0;
    u64 tile;
    { const u64 nt = ntiles, b = blockIdx.x, per = nt / 8, rem = nt % 8, x = b % 8, k = b / 8; tile = x * per + (x < rem ? x : rem) + k; }
    const u64 q = tile * (128ull << wl) + j * 16;
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
    u32 diff = (acc.x | acc.y | acc.z | acc.w) != 0;
    __syncthreads();
    if (diff) atomicOr(&D[j & 63], 1u << (tid & 31));
    __syncthreads();
    if (tid < 2) V[tile * 2 + tid] = D[tid] | ((u64)D[tid + 2] << 32);
    if (ntiles == 0) pad[tid] = 0;
}

// nine aligned 16-byte chunks per row (what a byte-unaligned 128-byte piece costs when it is fetched as aligned chunks):
// 56 row groups x 9 chunks = 504 threads, 896 rows; useful bytes = 896 rows x 128 B per tile
__global__ void __launch_bounds__(512, 4) k_skel9(const uint8_t* __restrict__ f, const u64* __restrict__ row_start, u64 ntiles, u64* __restrict__ V)
{
    extern __shared__ uint8_t pad[];
    __shared__ u32 D[64];
    const u32 tid = threadIdx.x, j = tid % 9u, sub = tid / 9u;
    if (tid < 64) D[tid] = 0;
    u64 tile;
    { const u64 nt = ntiles, b = blockIdx.x, per = nt / 8, rem = nt % 8, x = b % 8, k = b / 8; tile = x * per + (x < rem ? x : rem) + k; }
    const u64 q = tile * 128ull + j * 16;
    uint4 acc = make_uint4(0, 0, 0, 0);
    if (sub < 56) {
        const uint4 ref = load16u(f + row_start[0] + q);
        const ulonglong2* rp = reinterpret_cast<const ulonglong2*>(row_start + sub * 16u);
        ulonglong2 rv[8];
#pragma unroll
        for (int i = 0; i < 8; i++) rv[i] = rp[i];
        uint4 d[16];
#pragma unroll
        for (int i = 0; i < 8; i++) { d[2 * i] = load16u(f + rv[i].x + q); d[2 * i + 1] = load16u(f + rv[i].y + q); }
#pragma unroll
        for (int it = 0; it < 16; it++) { acc.x |= d[it].x ^ ref.x; acc.y |= d[it].y ^ ref.y; acc.z |= d[it].z ^ ref.z; acc.w |= d[it].w ^ ref.w; }
    }
    u32 diff = (acc.x | acc.y | acc.z | acc.w) != 0;
    __syncthreads();
    if (diff) atomicOr(&D[j & 63], 1u << (tid & 31));
    __syncthreads();
    if (tid < 2) V[tile * 2 + tid] = D[tid] | ((u64)D[tid + 2] << 32);
    if (ntiles == 0) pad[tid] = 0;
}

int main()
{
    const u64 S = 1000, L = 20000000, OCC2 = 64 * 1024;
    const u64 stride_u = L + 7, stride_a = L + 128;     // L % 128 == 0
    const u64 nbytes = 16 + S * (L + 256) + (1 << 20);
    uint8_t* f; CK(hipMalloc(&f, nbytes));
    k_fillw<<<4096, 256>>>((u32*)f, nbytes / 4);
    u64 *V, *d_rs; CK(hipMalloc(&V, (L / 64 + 64) * 8)); CK(hipMalloc(&d_rs, 1040 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute((const void*)k_skel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)OCC2));
    CK(hipFuncSetAttribute((const void*)k_linear, hipFuncAttributeMaxDynamicSharedMemorySize, (int)OCC2));
    auto timeit = [&](const char* name, auto launch) {
        launch(); CK(hipDeviceSynchronize());
        double best = 1e30;
        for (int i = 0; i < 3; i++) { CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = best < ms ? best : ms; }
        printf("%-64s %8.3f ms  %6.2f TB/s\n", name, best, S * L / best / 1e9); fflush(stdout);
    };
    auto set_rows = [&](u64 off0, u64 stride, u64 nrows) {
        std::vector<u64> rs(1040);
        for (u64 r = 0; r < 1040; r++) rs[r] = off0 + (r < nrows ? r : nrows - 1) * stride;
        CK(hipMemcpy(d_rs, rs.data(), 1040 * 8, hipMemcpyHostToDevice));
    };
    timeit("LIN  linear, 128-byte aligned", [&] { k_linear<<<(unsigned)(S * L / 131072), 512, OCC2>>>(f, V); });
    timeit("LINu linear, base + 5 bytes", [&] { k_linear<<<(unsigned)(S * L / 131072), 512, OCC2>>>(f + 5, V); });
    set_rows(5, stride_u, S);
    timeit("A     1000 rows x 128 B, byte-unaligned rows", [&] { k_skel<<<(unsigned)(L / 128), 512, OCC2>>>(f, d_rs, 0, L / 128, V); });
    set_rows(0, stride_a, S);
    timeit("Aa128 1000 rows x 128 B, 128-byte aligned pieces", [&] { k_skel<<<(unsigned)(L / 128), 512, OCC2>>>(f, d_rs, 0, L / 128, V); });
    set_rows(16, stride_a, S);
    timeit("Aa16  1000 rows x 128 B, 16-byte aligned pieces (offset 16)", [&] { k_skel<<<(unsigned)(L / 128), 512, OCC2>>>(f, d_rs, 0, L / 128, V); });
    set_rows(64, stride_a, S);
    timeit("Aa64  1000 rows x 128 B, 64-byte aligned pieces (offset 64)", [&] { k_skel<<<(unsigned)(L / 128), 512, OCC2>>>(f, d_rs, 0, L / 128, V); });
    CK(hipFuncSetAttribute((const void*)k_skel9, hipFuncAttributeMaxDynamicSharedMemorySize, (int)OCC2));
    for (u64 off : {16ull, 4ull, 5ull}) {
        char nm[128];
        set_rows(off, stride_a, S);
        snprintf(nm, sizeof nm, "A9    896 rows x 9 chunks of 16 B (144 B), row offset %llu (TB/s of the 128 useful bytes, x 0.896 rows)", off);
        timeit(nm, [&] { k_skel9<<<(unsigned)(L / 128), 512, OCC2>>>(f, d_rs, L / 128, V); });
        snprintf(nm, sizeof nm, "A     1000 rows x 128 B, row offset %llu", off);
        timeit(nm, [&] { k_skel<<<(unsigned)(L / 128), 512, OCC2>>>(f, d_rs, 0, L / 128, V); });
    }
    for (u32 wl = 1; wl <= 1; wl++) {
        const u64 rows = S >> wl, Lw = L << wl;          // same bytes: fewer, longer rows
        char nm[128];
        set_rows(5, Lw + 7, rows);
        snprintf(nm, sizeof nm, "W%u    %llu rows x %u B, byte-unaligned", 1u << wl, rows, 128u << wl);
        timeit(nm, [&] { k_skel<<<(unsigned)(Lw / (128ull << wl)), 512, OCC2>>>(f, d_rs, wl, Lw / (128ull << wl), V); });
        set_rows(0, Lw + 128, rows);
        snprintf(nm, sizeof nm, "W%ua   %llu rows x %u B, 128-byte aligned", 1u << wl, rows, 128u << wl);
        timeit(nm, [&] { k_skel<<<(unsigned)(Lw / (128ull << wl)), 512, OCC2>>>(f, d_rs, wl, Lw / (128ull << wl), V); });
    }
    return 0;
}
