// scan_skel2.hip — round-3 experiment 2: where does the register skeleton of k_scan_extract lose its 23 % against the
// per-CU HBM rate (18.6 of ~24 GB/s per CU)?  Variants of the same loads + mask reduction:
//
//   LIN      linear stream of the same bytes (ceiling on this box)
//   A<W>     one workgroup per tile, W waves per SIMD (4: two workgroups per CU as shipped; 6: three, <= 80 VGPRs)
//   AP<W>    persistent workgroups (row table in LDS, no dispatch / prologue between tiles)
//   AT<n>    A4 + "touch ahead": behind its own loads every wave issues LDS-DMA loads of the tile n tiles ahead into a
//            dummy LDS page (no registers): the lines are on their way to L2 / Infinity Cache while this workgroup is in
//            its tail and while the slot turns around
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o scan_skel2 scan_skel2.hip && ./scan_skel2 [S] [L] [reps]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef unsigned long long u64;
typedef unsigned int u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
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

// ---------------------------------------------------------------- LIN
__global__ void __launch_bounds__(512, 4) k_linear(const uint8_t* __restrict__ f, u64* __restrict__ out)
{
    const uint8_t* base = f + (u64)blockIdx.x * 131072 + threadIdx.x * 16;
    uint4 d[16];
#pragma unroll
    for (int i = 0; i < 16; i++) d[i] = *reinterpret_cast<const uint4*>(base + i * 8192);
    u32 a = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) a |= d[i].x ^ d[i].y ^ d[i].z ^ d[i].w;
    if (a == 0x12345678u) out[blockIdx.x] = a;
}

// ---------------------------------------------------------------- A / AP
template <int MINW, bool PERSIST>
__global__ void __launch_bounds__(512, MINW) k_skel(const uint8_t* __restrict__ f, const u64* __restrict__ row_start, u32 S,
                                                    u64 ntiles, u64* __restrict__ V)
{
    extern __shared__ uint8_t occupancy_pad[];                            // dynamic LDS only limits the workgroups per CU
    __shared__ u32 D[2][8];
    __shared__ __attribute__((aligned(16))) u64 rs[PERSIST ? 1024 : 2];
    const u32 tid = threadIdx.x, j = tid & 7, sub = tid >> 3;
    if (tid < 16) D[tid >> 3][tid & 7] = 0;
    if (ntiles == 0) occupancy_pad[tid] = 0;
    if constexpr (PERSIST) {
        for (u32 r = tid; r < 1024; r += 512) rs[r] = row_start[r];      // padded table (rows past S repeat the last one)
        __syncthreads();
    }
    const u64 x = blockIdx.x % 8, per = ntiles / 8, rem = ntiles % 8;
    const u64 t_begin = x * per + (x < rem ? x : rem), t_end = t_begin + per + (x < rem ? 1 : 0);
    const u64 wg = blockIdx.x / 8, nwg = PERSIST ? gridDim.x / 8 : 1;
    u32 cur = 0;
    for (u64 tile = t_begin + wg; tile < t_end; tile += nwg, cur ^= 1) {
        const u64 q = tile * 128 + j * 16;
        uint4 d[16];
        uint4 ref;
        if constexpr (PERSIST) {
            ref = load16u(f + rs[0] + q);
            const ulonglong2* rp = reinterpret_cast<const ulonglong2*>(rs + sub * 16u);
#pragma unroll
            for (int i = 0; i < 8; i++) { const ulonglong2 rv = rp[i]; d[2 * i] = load16u(f + rv.x + q); d[2 * i + 1] = load16u(f + rv.y + q); }
        } else {
            ref = load16u(f + row_start[0] + q);
            const ulonglong2* rp = reinterpret_cast<const ulonglong2*>(row_start + sub * 16u);
            ulonglong2 rv[8];
#pragma unroll
            for (int i = 0; i < 8; i++) rv[i] = rp[i];
#pragma unroll
            for (int i = 0; i < 8; i++) { d[2 * i] = load16u(f + rv[i].x + q); d[2 * i + 1] = load16u(f + rv[i].y + q); }
        }
        uint4 acc = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int it = 0; it < 16; it++) { acc.x |= d[it].x ^ ref.x; acc.y |= d[it].y ^ ref.y; acc.z |= d[it].z ^ ref.z; acc.w |= d[it].w ^ ref.w; }
        u32 diff = nz16(acc.x, acc.y, acc.z, acc.w);
        for (u32 o = 8; o < 64u; o <<= 1) diff |= (u32)__shfl_xor((int)diff, (int)o, 64);
        __syncthreads();                                                  // D[cur] is zero
        if ((tid & 63u) < 8 && diff) atomicOr(&D[cur][j], diff);
        __syncthreads();
        if (tid < 2) {
            u32* dd = D[cur];
            V[tile * 2 + tid] = (u64)dd[4 * tid] | ((u64)dd[4 * tid + 1] << 16) | ((u64)dd[4 * tid + 2] << 32) | ((u64)dd[4 * tid + 3] << 48);
            dd[4 * tid] = 0; dd[4 * tid + 1] = 0; dd[4 * tid + 2] = 0; dd[4 * tid + 3] = 0;
        }
        if constexpr (!PERSIST) break;
    }
}

// ---------------------------------------------------------------- AT: touch ahead
__device__ __forceinline__ u32x4 asm_load16(const uint8_t* p)
{
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void asm_touch(const uint8_t* g, u32 lds_dst)
{
    u32 keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds_dst) : "memory");
}
template <int NT /* touches per wave: 0, 8 (every other row) or 16 */>
__global__ void __launch_bounds__(512, 4) k_skel_touch(const uint8_t* __restrict__ f, const u64* __restrict__ row_start, u32 S,
                                                       u64 ntiles, u64* __restrict__ V, u32 ahead)
{
    extern __shared__ uint8_t occupancy_pad[];
    __shared__ u32 D[8];
    __shared__ __attribute__((aligned(1024))) uint8_t dummy[8 * 1024];
    const u32 tid = threadIdx.x, j = tid & 7, sub = tid >> 3;
    if (tid < 8) D[tid] = 0;
    if (ntiles == 0) occupancy_pad[tid] = 0;
    const u64 nt = ntiles, b = blockIdx.x, per = nt / 8, rem = nt % 8, x = b % 8, k = b / 8;
    const u64 t_begin = x * per + (x < rem ? x : rem), t_end = t_begin + per + (x < rem ? 1 : 0);
    const u64 tile = t_begin + k;
    const u64 q = tile * 128 + j * 16;
    const ulonglong2* rp = reinterpret_cast<const ulonglong2*>(row_start + sub * 16u);
    ulonglong2 rv[8];
#pragma unroll
    for (int i = 0; i < 8; i++) rv[i] = rp[i];
    const u64 r0 = row_start[0];
    const uint8_t* a[16];
#pragma unroll
    for (int i = 0; i < 8; i++) { a[2 * i] = f + rv[i].x + q; a[2 * i + 1] = f + rv[i].y + q; }
    u32x4 d[16];
    u32x4 ref = asm_load16(f + r0 + q);
#pragma unroll
    for (int i = 0; i < 16; i++) d[i] = asm_load16(a[i]);
    const bool touch = NT > 0 && tile + ahead < t_end;                    // workgroup-uniform
    if (touch) {
        const u32 dst = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(uintptr_t)(__attribute__((address_space(3))) void*)(dummy + (tid >> 6) * 1024));
        const u64 adv = (u64)ahead * 128;
#pragma unroll
        for (int i = 0; i < 16; i += 16 / (NT ? NT : 1)) asm_touch(a[i] + adv, dst);
    }
    // own loads are older than the touches: wait until only the touches are outstanding
#define DD(i) "+v"(d[i])
    if (touch) {
        if (NT == 16) asm volatile("s_waitcnt vmcnt(16)" : "+v"(ref), DD(0), DD(1), DD(2), DD(3), DD(4), DD(5), DD(6), DD(7), DD(8), DD(9), DD(10), DD(11), DD(12), DD(13), DD(14), DD(15) :: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" : "+v"(ref), DD(0), DD(1), DD(2), DD(3), DD(4), DD(5), DD(6), DD(7), DD(8), DD(9), DD(10), DD(11), DD(12), DD(13), DD(14), DD(15) :: "memory");
    } else asm volatile("s_waitcnt vmcnt(0)" : "+v"(ref), DD(0), DD(1), DD(2), DD(3), DD(4), DD(5), DD(6), DD(7), DD(8), DD(9), DD(10), DD(11), DD(12), DD(13), DD(14), DD(15) :: "memory");
#undef DD
    u32x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int it = 0; it < 16; it++) acc |= d[it] ^ ref;
    u32 diff = nz16(acc.x, acc.y, acc.z, acc.w);
    for (u32 o = 8; o < 64u; o <<= 1) diff |= (u32)__shfl_xor((int)diff, (int)o, 64);
    __syncthreads();
    if ((tid & 63u) < 8 && diff) atomicOr(&D[j], diff);
    __syncthreads();
    if (tid < 2) V[tile * 2 + tid] = (u64)D[4 * tid] | ((u64)D[4 * tid + 1] << 16) | ((u64)D[4 * tid + 2] << 32) | ((u64)D[4 * tid + 3] << 48);
}

static double time_ms(hipEvent_t a, hipEvent_t b) { float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms; }

int main(int argc, char** argv)
{
    const u64 S = argc > 1 ? strtoull(argv[1], 0, 10) : 1000;
    const u64 L = argc > 2 ? strtoull(argv[2], 0, 10) : 20000000;       // multiple of 128
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
    u64 *Vref, *V; CK(hipMalloc(&Vref, nwords * 8)); CK(hipMalloc(&V, nwords * 8));
    k_ref_mask<<<(unsigned)((L + 255) / 256), 256>>>(f, S, L, stride, off0, Vref);
    CK(hipDeviceSynchronize());
    std::vector<u64> href(nwords), hv(nwords);
    CK(hipMemcpy(href.data(), Vref, nwords * 8, hipMemcpyDeviceToHost));
    printf("S=%llu L=%llu bytes=%.2f GB\n", S, L, S * L / 1e9);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto check = [&](const char* name, double ms, bool has_mask) {
        u64 bad = 0;
        if (has_mask) { CK(hipMemcpy(hv.data(), V, nwords * 8, hipMemcpyDeviceToHost)); for (u64 i = 0; i < nwords; i++) bad += hv[i] != href[i]; }
        printf("%-50s %8.3f ms  %6.2f TB/s  %5.1f GB/s/CU  %s\n", name, ms, S * L / ms / 1e9, S * L / ms / 1e6 / 256,
               !has_mask ? "" : bad ? "mask WRONG" : "mask ok");
        fflush(stdout);
        CK(hipMemset(V, 0, nwords * 8));
    };
    auto run = [&](const char* name, bool has_mask, auto launch) {
        launch(); CK(hipDeviceSynchronize());
        double best = 1e30;
        for (int i = 0; i < reps; i++) { CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); best = std::min(best, time_ms(e0, e1)); }
        check(name, best, has_mask);
    };
    const u64 ntiles = L / 128;
    run("LIN linear stream, 128 KB per workgroup", false, [&] { k_linear<<<(unsigned)(S * L / 131072), 512>>>(f, V); });
    if (S <= 1008) {
        const size_t OCC2 = 64 * 1024, OCC3 = 44 * 1024, OCC1 = 100 * 1024;       // dynamic LDS that leaves 2 / 3 / 1 workgroups per CU
        for (auto kp : {(const void*)k_skel<4, false>, (const void*)k_skel<6, false>, (const void*)k_skel<4, true>, (const void*)k_skel<6, true>,
                        (const void*)k_skel_touch<0>, (const void*)k_skel_touch<8>, (const void*)k_skel_touch<16>})
            CK(hipFuncSetAttribute(kp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)OCC1));
        run("A   one workgroup per tile, 1 per CU", true, [&] { k_skel<6, false><<<(unsigned)ntiles, 512, OCC1>>>(f, d_rs, (u32)S, ntiles, V); });
        run("A   one workgroup per tile, 2 per CU", true, [&] { k_skel<6, false><<<(unsigned)ntiles, 512, OCC2>>>(f, d_rs, (u32)S, ntiles, V); });
        run("A   one workgroup per tile, 3 per CU", true, [&] { k_skel<6, false><<<(unsigned)ntiles, 512, OCC3>>>(f, d_rs, (u32)S, ntiles, V); });
        run("AP  persistent, 2 per CU", true, [&] { k_skel<6, true><<<512, 512, OCC2>>>(f, d_rs, (u32)S, ntiles, V); });
        run("AP  persistent, 3 per CU", true, [&] { k_skel<6, true><<<768, 512, OCC3>>>(f, d_rs, (u32)S, ntiles, V); });
        run("AT0 asm loads, no touch, 2 per CU (control)", true, [&] { k_skel_touch<0><<<(unsigned)ntiles, 512, OCC2>>>(f, d_rs, (u32)S, ntiles, V, 0); });
        for (u32 ahead : {16u, 64u, 128u, 512u}) {
            char nm[96];
            snprintf(nm, sizeof nm, "AT16 2 per CU, touch 16/16 row groups, %u tiles ahead", ahead);
            run(nm, true, [&] { k_skel_touch<16><<<(unsigned)ntiles, 512, OCC2>>>(f, d_rs, (u32)S, ntiles, V, ahead); });
        }
        for (u32 ahead : {64u, 128u}) {
            char nm[96];
            snprintf(nm, sizeof nm, "AT8  2 per CU, touch 8/16 row groups, %u tiles ahead", ahead);
            run(nm, true, [&] { k_skel_touch<8><<<(unsigned)ntiles, 512, OCC2>>>(f, d_rs, (u32)S, ntiles, V, ahead); });
        }
    }
    return 0;
}
