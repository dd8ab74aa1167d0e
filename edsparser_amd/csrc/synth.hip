// synth.hip — genrandomeds-shaped synthetic alignment generated directly in HBM.
//
// Shape follows the reference's generator (src/cpp/tools/genrandomeds.cpp:221-352): reference
// uniform over ACGT, a fraction v of the columns are variant sites, each site has k ~ U[2,4]
// alternatives (alternative 0 = reference base; the others 70 % SNP to a different base, else
// 50/50 insertion of 1..10 random bases after the reference base / deletion), path p < k takes
// alternative p, the other paths choose uniformly.  Rows = paths; insertions become gap-padded
// columns, deletions '-'.  Counter-based: every byte depends only on (seed, global column, row),
// so any column slab generated on any GPU matches the same columns of the whole alignment.
#include "synth.hpp"

namespace edsx {

__host__ __device__ static inline u64 cumdigits(u64 r)
{   // sum of decimal digit counts of 0..r-1
    u64 t = r;
    for (u64 p = 10; p <= r; p *= 10) { t += r - p; if (p > r / 10) break; }
    return t;
}
__host__ __device__ static inline u32 ndig_host(u64 v) { u32 d = 1; while (v >= 10) { v /= 10; d++; } return d; }

size_t synth_size(u32 S, u64 ncols)
{
    return (size_t)S * (ncols + 1 + 3) + cumdigits(S);
}

// Row-aligned variant: every header is padded with blanks so that the row's first column is a multiple of `align` bytes
// from the image (what an upload that lays the rows out for the column scan produces; the FASTA stays valid).
// Header bytes of row 0 / of every later row (incl. '>' and '\n'):
__host__ __device__ static inline u64 aligned_hdr0(u64 align) { return align * ((8 + align - 1) / align); }
__host__ __device__ static inline u64 aligned_hdr(u64 ncols, u64 align)
{
    u64 h = (align - (ncols + 1) % align) % align;
    while (h < 8) h += align;                       // room for ">s<idx>" of up to five digits and the newline
    return h;
}
size_t synth_size_aligned(u32 S, u64 ncols, u32 align)
{
    if (align <= 1) return synth_size(S, ncols);
    return (size_t)(aligned_hdr0(align) + (u64)S * (ncols + 1) + (u64)(S - 1) * aligned_hdr(ncols, align));
}

__device__ __forceinline__ bool site_raw(u64 seed, u64 gc, u64 vthr) { return (hash3(seed, gc, 1) >> 40) < vthr; }

struct SiteAlts { u32 k; u32 kind[4]; u32 m[4]; u32 base[4]; };   // kind: 0 ref, 1 SNP, 2 INS, 3 DEL
__device__ SiteAlts site_alts(u64 seed, u64 gc)
{
    SiteAlts s;
    const u32 refb = (u32)hash3(seed, gc, 3) & 3u;
    s.k = 2 + (u32)(hash3(seed, gc, 2) % 3);
    s.kind[0] = 0; s.m[0] = 0; s.base[0] = refb;
    for (u32 a = 1; a < 4; a++) {
        u64 t = hash3(seed, gc, 16 + a);
        u32 u = (u32)t & 0xffffu;
        s.m[a] = 0; s.base[a] = refb;
        if (a >= s.k) { s.kind[a] = 0; continue; }
        if (u < 45875u) { s.kind[a] = 1; s.base[a] = (refb + 1 + (u32)((t >> 16) % 3)) & 3u; }
        else if ((t >> 20) & 1) { s.kind[a] = 2; s.m[a] = 1 + (u32)((t >> 24) % 10); }
        else s.kind[a] = 3;
    }
    return s;
}

// one 64-bit descriptor per column: k | ch[0..3] << 8.. | owner distance << 40
__global__ void k_synth_desc(u64* __restrict__ desc, u64 col0, u64 ncols, u64 seed, u64 vthr)
{
    const char* ACGT = "ACGT";
    for (u64 c = blockIdx.x * (u64)blockDim.x + threadIdx.x; c < ncols; c += (u64)gridDim.x * blockDim.x) {
        const u64 gc = col0 + c;
        const u32 refb = (u32)hash3(seed, gc, 3) & 3u;
        u64 d = 0;
        bool done = false;
        for (u32 j = 1; j <= 10 && !done; j++) {
            if (gc < j || !site_raw(seed, gc - j, vthr)) continue;
            SiteAlts s = site_alts(seed, gc - j);
            u32 maxm = 0;
            for (u32 a = 1; a < s.k; a++) if (s.kind[a] == 2 && s.m[a] > maxm) maxm = s.m[a];
            if (maxm < j) continue;
            d = s.k | ((u64)j << 40);
            for (u32 a = 0; a < 4; a++) {
                u32 ch = '-';
                if (a < s.k && s.kind[a] == 2 && s.m[a] >= j) ch = ACGT[hash3(seed, gc - j, 64 + a * 16 + j) & 3];
                d |= (u64)ch << (8 + 8 * a);
            }
            done = true;
        }
        if (!done) {
            if (site_raw(seed, gc, vthr)) {
                SiteAlts s = site_alts(seed, gc);
                d = s.k;
                for (u32 a = 0; a < 4; a++) {
                    u32 ch = ACGT[refb];
                    if (a < s.k) {
                        if (s.kind[a] == 1) ch = ACGT[s.base[a]];
                        else if (s.kind[a] == 3) ch = '-';
                    }
                    d |= (u64)ch << (8 + 8 * a);
                }
            } else {
                d = (u64)ACGT[refb] << 8;
            }
        }
        desc[c] = d;
    }
}

struct __attribute__((packed, aligned(1))) Pack16 { uint8_t b[16]; };

__global__ void k_synth_fill(uint8_t* __restrict__ out, const u64* __restrict__ desc, u32 S, u64 col0,
                             u64 ncols, u64 seed, u32 align)
{
    const u64 chunks = (ncols + 15) / 16;
    const u64 total = (u64)S * chunks;
    for (u64 t = blockIdx.x * (u64)blockDim.x + threadIdx.x; t < total; t += (u64)gridDim.x * blockDim.x) {
        const u32 row = (u32)(t / chunks);
        const u64 ch = t % chunks;
        const u32 nd = ndig_host(row);
        u64 rowbase, hlen;                                                   // offset of '>', header bytes incl. the newline
        if (align <= 1) { rowbase = (u64)row * (ncols + 1 + 3) + cumdigits(row); hlen = 2 + nd + 1; }
        else {
            const u64 h0 = aligned_hdr0(align), h = aligned_hdr(ncols, align);
            rowbase = row ? h0 + (ncols + 1) + (u64)(row - 1) * (h + ncols + 1) : 0;
            hlen = row ? h : h0;
        }
        uint8_t* data = out + rowbase + hlen;
        if (ch == 0) {
            out[rowbase] = '>'; out[rowbase + 1] = 's';
            u32 v = row;
            for (int i = (int)nd - 1; i >= 0; i--) { out[rowbase + 2 + i] = (uint8_t)('0' + v % 10); v /= 10; }
            for (u64 i = 2 + nd; i + 1 < hlen; i++) out[rowbase + i] = ' ';
            out[rowbase + hlen - 1] = '\n';
            data[ncols] = '\n';
        }
        Pack16 pk;
        const u64 c0 = ch * 16;
        const int n = (ncols - c0) < 16 ? (int)(ncols - c0) : 16;
        for (int i = 0; i < n; i++) {
            const u64 d = desc[c0 + i];
            const u32 k = (u32)d & 0xffu;
            u32 choice = 0;
            if (k) {
                const u64 owner = col0 + c0 + i - ((d >> 40) & 0xff);
                choice = row < k ? row : (u32)(hash3(seed ^ 0xA5A5A5A5ull, owner, row) % k);
            }
            pk.b[i] = (uint8_t)(d >> (8 + 8 * choice));
        }
        if (n == 16) __builtin_memcpy(data + c0, &pk, 16);
        else for (int i = 0; i < n; i++) data[c0 + i] = pk.b[i];
    }
}

void synth_generate(uint8_t* d_out, u64* d_desc, u32 S, u64 col0, u64 ncols, double v, u64 seed, hipStream_t st, u32 align)
{
    u64 vthr = (u64)(v * 16777216.0);
    hipLaunchKernelGGL(k_synth_desc, dim3(4096), dim3(256), 0, st, d_desc, col0, ncols, seed, vthr);
    hipLaunchKernelGGL(k_synth_fill, dim3(8192), dim3(256), 0, st, d_out, d_desc, S, col0, ncols, seed, align);
}

} // namespace edsx
