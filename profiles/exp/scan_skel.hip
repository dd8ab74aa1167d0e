// scan_skel.hip — round-3 experiment: which load structure streams "S rows x W-byte pieces" closest to the
// per-CU HBM ceiling (~24 GB/s per CU)?  Stand-alone (no libedsx): builds a synthetic one-line-per-row image,
// computes the variant-column mask with several kernels, checks them against a plain kernel, times them.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o scan_skel scan_skel.hip && ./scan_skel [S] [L] [reps]
//
//   T   does global_load_lds_dwordx4 accept byte-unaligned source addresses?
//   A   register skeleton of k_scan_extract (512 threads, thread = 16 rows x 16 B, 2 workgroups per CU)
//   B   LDS-DMA, persistent workgroup per CU, double-buffered tiles (S x PW bytes each), mask from LDS
//   D   LDS-DMA, one workgroup per tile (dispatcher back-fills), 2 workgroups per CU
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
        const u64 r = (i * 4) / L, c = (i * 4) % L;      // L % 4 == 0
        uint8_t* p = f + off0 + r * stride + c;
        p[0] = cell(r, c); p[1] = cell(r, c + 1); p[2] = cell(r, c + 2); p[3] = cell(r, c + 3);
    }
}
// plain check kernel: thread per column
__global__ void k_ref_mask(const uint8_t* f, u64 S, u64 L, u64 stride, u64 off0, u64* V)
{
    const u64 c = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    bool var = false;
    if (c < L) { const uint8_t r0 = f[off0 + c]; for (u64 r = 1; r < S; r++) var |= f[off0 + r * stride + c] != r0; }
    const u64 b = __ballot(var);
    if ((threadIdx.x & 63) == 0 && c < L) V[c >> 6] = b;
}

__device__ __forceinline__ uint4 load16u(const uint8_t* p) { uint4 v; __builtin_memcpy(&v, p, 16); return v; }
__device__ __forceinline__ u32 ne_bytes4(u32 a, u32 b)
{
    u32 x = a ^ b;
    u32 h = (((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u;
    return ((h >> 7) | (h >> 14) | (h >> 21) | (h >> 28)) & 0xfu;
}
__device__ __forceinline__ u32 chunk_ne16(const uint4& a, const uint4& b)
{
    return ne_bytes4(a.x, b.x) | (ne_bytes4(a.y, b.y) << 4) | (ne_bytes4(a.z, b.z) << 8) | (ne_bytes4(a.w, b.w) << 12);
}
__device__ __forceinline__ u32 nz_bytes16(const uint4& a) { return chunk_ne16(a, make_uint4(0, 0, 0, 0)); }

// ---------------------------------------------------------------- T: alignment of the LDS-DMA source
__global__ void k_dma_align(const uint8_t* src, uint8_t* out)
{
    __shared__ __attribute__((aligned(16))) uint8_t buf[1024];
    for (int off = 0; off < 16; off++) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + off + threadIdx.x * 16),
                                         (__attribute__((address_space(3))) void*)buf, 16, 0, 0);
        __syncthreads();
        for (int i = threadIdx.x; i < 1024; i += 64) out[off * 1024 + i] = buf[i];
        __syncthreads();
    }
}

// ---------------------------------------------------------------- A: register skeleton (as k_scan_extract's loads + reduce)
__global__ void __launch_bounds__(512, 4) k_skel_reg(const uint8_t* __restrict__ f, const u64* __restrict__ row_start, u32 S, u64 L,
                                                     u64 ntiles, u64* __restrict__ V, int transpose)
{
    __shared__ u32 D[8];
    __shared__ __attribute__((aligned(16))) uint8_t colbuf[40960];
    const u32 tid = threadIdx.x, j = tid & 7, sub = tid >> 3;
    u64 tile;
    { const u64 nt = ntiles, b = blockIdx.x, per = nt / 8, rem = nt % 8, x = b % 8, k = b / 8; tile = x * per + (x < rem ? x : rem) + k; }
    const u64 q = tile * 128 + j * 16;
    if (tid < 8) D[tid] = 0;
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
    u32 diff = nz_bytes16(acc);
    for (u32 o = 8; o < 64u; o <<= 1) diff |= (u32)__shfl_xor((int)diff, (int)o, 64);
    __syncthreads();
    if ((tid & 63u) < 8 && diff) atomicOr(&D[j], diff);
    __syncthreads();
    if (transpose) {            // the 128 v_perm of the real kernel + the LDS image of the variant columns
        const u32 V16 = D[j];
        u32 pre = 0;
        for (u32 c = 0; c < j; c++) pre += __builtin_popcount(D[c]);
        uint32_t tr[16][4];
#define EDSX_T(C, COMP)                                                                            \
        _Pragma("unroll") for (int k4 = 0; k4 < 4; k4++) {                                        \
            const uint32_t a0 = d[4 * k4].COMP, a1 = d[4 * k4 + 1].COMP, a2 = d[4 * k4 + 2].COMP, a3 = d[4 * k4 + 3].COMP; \
            const uint32_t t0 = __builtin_amdgcn_perm(a1, a0, 0x05010400u), t1 = __builtin_amdgcn_perm(a1, a0, 0x07030602u); \
            const uint32_t t2 = __builtin_amdgcn_perm(a3, a2, 0x05010400u), t3 = __builtin_amdgcn_perm(a3, a2, 0x07030602u); \
            tr[4 * C][k4] = __builtin_amdgcn_perm(t2, t0, 0x05040100u);                           \
            tr[4 * C + 1][k4] = __builtin_amdgcn_perm(t2, t0, 0x07060302u);                       \
            tr[4 * C + 2][k4] = __builtin_amdgcn_perm(t3, t1, 0x05040100u);                       \
            tr[4 * C + 3][k4] = __builtin_amdgcn_perm(t3, t1, 0x07060302u);                       \
        }
        EDSX_T(0, x) EDSX_T(1, y) EDSX_T(2, z) EDSX_T(3, w)
#undef EDSX_T
        if (V16 && pre < 36) {
            uint8_t* dst = colbuf + (size_t)pre * 1024 + sub * 16u;
#define EDSX_L(I) if ((V16 & (1u << I)) && dst < colbuf + 39 * 1024) { *reinterpret_cast<uint4*>(dst) = make_uint4(tr[I][0], tr[I][1], tr[I][2], tr[I][3]); dst += 1024; }
            EDSX_L(0) EDSX_L(1) EDSX_L(2) EDSX_L(3) EDSX_L(4) EDSX_L(5) EDSX_L(6) EDSX_L(7)
            EDSX_L(8) EDSX_L(9) EDSX_L(10) EDSX_L(11) EDSX_L(12) EDSX_L(13) EDSX_L(14) EDSX_L(15)
#undef EDSX_L
        }
        __syncthreads();
        if (tid == 0 && colbuf[17] == 1) D[0] ^= 0;   // keep the image alive
    }
    if (tid < 2) V[tile * 2 + tid] = (u64)D[4 * tid] | ((u64)D[4 * tid + 1] << 16) | ((u64)D[4 * tid + 2] << 32) | ((u64)D[4 * tid + 3] << 48);
}

// ---------------------------------------------------------------- B / D: LDS-DMA tiles
// Tile = SP rows x PW bytes (PW = 64 or 128; LPR = PW/16 lanes per row).  One wave-instruction moves 64/LPR rows.
// LDS image: row r at r*PW, its 16-byte chunk c in slot c ^ swz(r) (source-side swizzle: the lane that writes slot s
// fetches chunk s ^ swz(r)), so that the thread-per-row ds_read_b128 of the mask phase is conflict-free.
template <int PW> __device__ __forceinline__ u32 swz(u32 r) { return (r >> (PW == 64 ? 2 : 1)) & (PW / 16 - 1); }

template <int PW, int T>
__device__ __forceinline__ void dma_tile(const uint8_t* __restrict__ f, const u64* rs /*LDS*/, u32 SP, u64 q0, uint8_t* buf)
{
    constexpr u32 LPR = PW / 16, RPI = 64 / LPR;          // lanes per row, rows per instruction
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (u32 r0 = wave * RPI; r0 < SP; r0 += (T / 64) * RPI) {
        const u32 r = r0 + lane / LPR, s = lane % LPR;
        const uint8_t* g = f + rs[r] + q0 + ((s ^ swz<PW>(r)) * 16u);
        // inline asm: hipcc would otherwise drain the DMA (vmcnt(0)) in front of every later LDS read
        const u32 dst = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(uintptr_t)(__attribute__((address_space(3))) void*)(buf + (size_t)r0 * PW));
        u32 keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
    }
}
// mask of the tile from its LDS image: thread per row (rows tid, tid + T, ..)
template <int PW, int T>
__device__ __forceinline__ void mask_tile(const uint8_t* buf, u32 S, u32* Vsh)
{
    constexpr u32 NCH = PW / 16;
    u32 m[NCH / 2];
#pragma unroll
    for (u32 i = 0; i < NCH / 2; i++) m[i] = 0;
    for (u32 r = threadIdx.x; r < S; r += T) {
#pragma unroll
        for (u32 c = 0; c < NCH; c++) {
            const uint4 ref = *reinterpret_cast<const uint4*>(buf + c * 16u);                    // row 0: swz 0, broadcast
            const uint4 v = *reinterpret_cast<const uint4*>(buf + (size_t)r * PW + ((c ^ swz<PW>(r)) * 16u));
            m[c >> 1] |= chunk_ne16(v, ref) << ((c & 1) * 16);
        }
    }
#pragma unroll
    for (u32 i = 0; i < NCH / 2; i++) {
        u32 v = m[i];
        for (int o = 32; o > 0; o >>= 1) v |= (u32)__shfl_xor((int)v, o, 64);
        if ((threadIdx.x & 63) == 0 && v) atomicOr(&Vsh[i], v);
    }
}

template <int PW, int T, bool PERSIST>
__global__ void __launch_bounds__(T) k_skel_dma(const uint8_t* __restrict__ f, const u64* __restrict__ row_start, u32 S, u32 SP,
                                                u64 ntiles, u64* __restrict__ V)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    u64* rs = reinterpret_cast<u64*>(lds);                                  // SP row starts
    u32* Vsh = reinterpret_cast<u32*>(lds + (size_t)SP * 8);                // 2 x 4 dwords
    uint8_t* buf0 = lds + (size_t)SP * 8 + 64;
    uint8_t* buf1 = buf0 + (size_t)SP * PW;
    for (u32 r = threadIdx.x; r < SP; r += T) rs[r] = row_start[r < S ? r : S - 1];
    if (threadIdx.x < 8) Vsh[threadIdx.x] = 0;
    __syncthreads();
    constexpr u32 NW = PW / 64;                                             // u64 mask words per tile
    if constexpr (!PERSIST) {
        u64 tile;
        { const u64 nt = ntiles, b = blockIdx.x, per = nt / 8, rem = nt % 8, x = b % 8, k = b / 8; tile = x * per + (x < rem ? x : rem) + k; }
        dma_tile<PW, T>(f, rs, SP, tile * PW, buf0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        mask_tile<PW, T>(buf0, S, Vsh);
        __syncthreads();
        if (threadIdx.x < NW) V[tile * NW + threadIdx.x] = (u64)Vsh[2 * threadIdx.x] | ((u64)Vsh[2 * threadIdx.x + 1] << 32);
    } else {
        // XCD x = blockIdx % 8 owns a contiguous range of tiles; its workgroups walk it side by side
        const u64 x = blockIdx.x % 8, wg = blockIdx.x / 8, nwg = gridDim.x / 8;
        const u64 per = ntiles / 8, rem = ntiles % 8;
        const u64 t_begin = x * per + (x < rem ? x : rem), t_end = t_begin + per + (x < rem ? 1 : 0);
        u64 t = t_begin + wg;
        if (t < t_end) dma_tile<PW, T>(f, rs, SP, t * PW, buf0);
        u32 cur = 0;
        for (; t < t_end; t += nwg, cur ^= 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // tile t has landed,
            __syncthreads();                                                // everyone is done with the other buffer
            if (t + nwg < t_end) dma_tile<PW, T>(f, rs, SP, (t + nwg) * PW, cur ? buf0 : buf1);
            mask_tile<PW, T>(cur ? buf1 : buf0, S, Vsh + 4 * cur);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (threadIdx.x < NW) {
                u32* vs = Vsh + 4 * cur;
                V[t * NW + threadIdx.x] = (u64)vs[2 * threadIdx.x] | ((u64)vs[2 * threadIdx.x + 1] << 32);
                vs[2 * threadIdx.x] = 0; vs[2 * threadIdx.x + 1] = 0;
            }
        }
    }
}

static double time_ms(hipEvent_t a, hipEvent_t b) { float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms; }

int main(int argc, char** argv)
{
    const u64 S = argc > 1 ? strtoull(argv[1], 0, 10) : 1000;
    const u64 L = argc > 2 ? strtoull(argv[2], 0, 10) : 20000000;       // multiple of 128
    const int reps = argc > 3 ? atoi(argv[3]) : 3;
    const u64 stride = L + 7, off0 = 5;
    const u64 nbytes = off0 + S * stride + 4096;
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
    u64 nvar = 0; for (u64 w : href) nvar += __builtin_popcountll(w);
    printf("S=%llu L=%llu bytes=%.2f GB variant columns=%.2f%%\n", S, L, S * L / 1e9, 100.0 * nvar / L);

    // ---- T
    {
        std::vector<uint8_t> src(2048), out(16 * 1024);
        for (int i = 0; i < 2048; i++) src[i] = (uint8_t)(i * 7 + (i >> 8));
        uint8_t *ds, *dout; CK(hipMalloc(&ds, 2048)); CK(hipMalloc(&dout, 16 * 1024));
        CK(hipMemcpy(ds, src.data(), 2048, hipMemcpyHostToDevice));
        k_dma_align<<<1, 64>>>(ds, dout);
        CK(hipMemcpy(out.data(), dout, 16 * 1024, hipMemcpyDeviceToHost));
        for (int off = 0; off < 16; off++) {
            int bad = 0;
            for (int i = 0; i < 1024; i++) bad += out[off * 1024 + i] != src[off + i];
            printf("T: LDS-DMA dwordx4, source offset %2d: %s (%d bytes differ)\n", off, bad ? "WRONG" : "ok", bad);
        }
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto check = [&](const char* name, double ms) {
        CK(hipMemcpy(hv.data(), V, nwords * 8, hipMemcpyDeviceToHost));
        u64 bad = 0; for (u64 i = 0; i < nwords; i++) bad += hv[i] != href[i];
        printf("%-44s %8.3f ms  %6.2f TB/s  %5.1f GB/s/CU  mask %s (%llu words differ)\n", name, ms, S * L / ms / 1e9, S * L / ms / 1e6 / 256,
               bad ? "WRONG" : "ok", bad);
        fflush(stdout);
        CK(hipMemset(V, 0, nwords * 8));
    };
    auto run = [&](const char* name, auto launch) {
        launch(); CK(hipDeviceSynchronize());
        double best = 1e30;
        for (int i = 0; i < reps; i++) { CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); best = std::min(best, time_ms(e0, e1)); }
        check(name, best);
    };
    // ---- A
    if (S <= 1008) {
        run("A  register skeleton (loads+reduce)", [&] { k_skel_reg<<<(unsigned)(L / 128), 512>>>(f, d_rs, (u32)S, L, L / 128, V, 0); });
        run("A' register skeleton + transposes + LDS image", [&] { k_skel_reg<<<(unsigned)(L / 128), 512>>>(f, d_rs, (u32)S, L, L / 128, V, 1); });
    }
    // ---- B / D
    auto dma = [&](auto kern, const char* name, int PW, int T, bool persist, int nbuf, unsigned grid_mult) {
        const u32 RPI = 64 / (PW / 16), SP = (u32)((S + RPI - 1) / RPI * RPI);
        const size_t lds = (size_t)SP * 8 + 64 + (size_t)nbuf * SP * PW;
        if (lds > 160 * 1024) { printf("%-44s skipped (LDS %zu)\n", name, lds); return; }
        CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const u64 ntiles = L / PW;
        const unsigned grid = persist ? 256 * grid_mult : (unsigned)ntiles;
        run(name, [&] { kern<<<grid, T, lds>>>(f, d_rs, (u32)S, SP, ntiles, V); });
    };
    dma(k_skel_dma<64, 1024, true>, "B  DMA persistent 64B pieces x2 buf, 1024 thr", 64, 1024, true, 2, 1);
    dma(k_skel_dma<64, 512, true>, "B  DMA persistent 64B pieces x2 buf, 512 thr", 64, 512, true, 2, 1);
    dma(k_skel_dma<128, 1024, true>, "B  DMA persistent 128B pieces x2 buf, 1024 thr", 128, 1024, true, 2, 1);
    dma(k_skel_dma<64, 1024, false>, "D  DMA per-tile WG 64B pieces, 1024 thr", 64, 1024, false, 1, 1);
    dma(k_skel_dma<64, 512, false>, "D  DMA per-tile WG 64B pieces, 512 thr", 64, 512, false, 1, 1);
    dma(k_skel_dma<128, 1024, false>, "D  DMA per-tile WG 128B pieces, 1024 thr", 128, 1024, false, 1, 1);
    dma(k_skel_dma<128, 512, false>, "D  DMA per-tile WG 128B pieces, 512 thr", 128, 512, false, 1, 1);
    return 0;
}
