// msa_device.hip — MSA -> EDS / l-EDS on gfx950 (MI355X).  Hand-written HIP, wave64, no MFMA:
// this is byte/bit work bounded by HBM bandwidth.
//
// Replaces the reference's three passes (src/cpp/lib/transforms/msa_transforms.cpp):
//   pass 1 parse_msa_and_build_variant_bv :36-90   -> k_find_*, k_index_spec/_check/_rows + k_scan_extract
//   pass 2 build_eds/leds_boundaries      :101-190 -> k_runstart_words .. k_write_segs
//   pass 3 generate_output                :200-324 -> k_seg_meta, k_seg_count_fast (wave per segment, S <= 1024)
//                                                     / k_seg_count (workgroup per segment), scans,
//                                                     k_emit_fast2 / k_emit_fast / k_emit_variant (+ the common text in front)
//
// Data layout in HBM
//   file image    the FASTA bytes as given (no repacking).  Row r's raw byte q lives at
//                 row_start[r] + q; raw position q of alignment column c is c + c/lw for wrapped
//                 rows (msa_transforms.cpp:268-269) and c for one-line rows.
//   Vraw / V      1 bit per raw position / alignment column: 1 = variant column (some row differs
//                 from row 0, or row 0 has '-').  Complement of the reference's bit-vector B.
//   vc            the variant columns only, column-major: vc[slot * Spad + r] = byte of row r
//                 (natural row order: a lane's 16-byte load at 16*lane holds rows 16*lane .. +15).
//                 Slots are handed out per tile by an atomic counter (unordered between tiles,
//                 consecutive inside a 64-column word); word_slot[w] = slot of the first variant
//                 column of raw word w.  Every later kernel reads rows through vc, so the S x L
//                 matrix is read from HBM exactly once.
//   run/seg table seg_start[nseg+1] (alignment columns), bitmap Hseg of segment starts with a
//                 per-word prefix segbase[]; a segment is common iff V[seg_start] == 0.
//   rec           one grouping record per variant segment (count -> emit): group id of every row (2 or 4
//                 bits, natural row order), number of strings, representative rows / the .eds text.
//   eds_off/seds_off  exclusive scans of the per-segment text sizes.
#include "msa_device.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace edsx {

// ---------------------------------------------------------------------------------------------
// device header block
// ---------------------------------------------------------------------------------------------
enum : u64 {
    ST_NOT_FASTA = 1, ST_LAYOUT = 2, ST_TOO_MANY_ROWS = 4, ST_VC_OVERFLOW = 8,
    ST_NEWLINE_IN_DATA = 16, ST_FEW_ROWS = 32
};

// tell the compiler a value is wave-uniform (it then lives in SGPRs and branches on it are scalar)
__device__ __forceinline__ u64 uniform64(u64 v)
{
    return ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)(v >> 32)) << 32) |
           (u32)__builtin_amdgcn_readfirstlane((int)(u32)v);
}
__device__ __forceinline__ u32 uniform32(u32 v) { return (u32)__builtin_amdgcn_readfirstlane((int)v); }

__device__ __forceinline__ u64 ld_relaxed(const u64* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------------------
// K0: row index.  msa_transforms.cpp:46-68 (header lines, start_positions, line_width).
// ---------------------------------------------------------------------------------------------
__global__ void k_find_hdr_end(const uint8_t* __restrict__ f, u64 n, MsaHdr* h)
{
    const u32 lane = threadIdx.x;
    h->first_nl = n;
    h->first_hdr2 = n;
    if (n == 0 || f[0] != '>') {
        if (lane == 0) { h->status |= ST_NOT_FASTA; h->hdr_end = n; }
        return;
    }
    u64 pos = n;
    for (u64 base = 0; base < n; base += 64) {
        u64 i = base + lane;
        u64 b = ballot64(i < n && f[i] == '\n');
        if (b) { pos = base + __builtin_ctzll(b); break; }
    }
    if (lane == 0) h->hdr_end = pos;
}

// first '\n' and first "\n>" after the first header; persistent grid, windows visited in order,
// workgroups stop as soon as a hit lies before their next window.
__global__ void __launch_bounds__(1024) k_find_row0(const uint8_t* __restrict__ f, u64 n, MsaHdr* h)
{
    const u64 hdr_end = h->hdr_end;
    if (hdr_end >= n) return;
    const u64 start0 = hdr_end + 1;
    for (u64 w = blockIdx.x;; w += gridDim.x) {
        u64 base = start0 + w * (u64)(1024 * 16);
        if (base >= n) break;
        if (base > ld_relaxed(&h->first_hdr2)) break;
        u64 p = base + (u64)threadIdx.x * 16;
        if (p >= n) continue;
        int nb = (n - p) < 16 ? (int)(n - p) : 16;
        uint4 v = nb == 16 ? load16u(f + p) : load_partial(f + p, nb);
        u32 nl = eq_byte4(v.x, 0x0a0a0a0au) | (eq_byte4(v.y, 0x0a0a0a0au) << 4) |
                 (eq_byte4(v.z, 0x0a0a0a0au) << 8) | (eq_byte4(v.w, 0x0a0a0a0au) << 12);
        nl &= (nb == 16) ? 0xffffu : ((1u << nb) - 1u);
        if (!nl) continue;
        u64 first = p + __builtin_ctz(nl);
        if (first < ld_relaxed(&h->first_nl)) atomicMin(&h->first_nl, first);
        while (nl) {
            int i = __builtin_ctz(nl);
            nl &= nl - 1;
            u64 q = p + i;
            if (q + 1 < n && f[q + 1] == '>') { atomicMin(&h->first_hdr2, q); break; }
        }
    }
}

// one wave: geometry + the chain of row starts (each row start depends on the previous header).
// ---- speculative parallel row index -------------------------------------------------------------
// The chain in k_index_rows costs one dependent HBM load per row (~0.65 us).  Rows have equal data
// length and headers of nearly equal length, so header r is close to r * (distance of the first two
// headers): one wave per row searches a window around that guess for "\n>" (radius 64 + 8 r bytes,
// less than half a row), k_index_check then verifies that the found headers chain EXACTLY as the
// serial walk would see them (every header starts right behind the previous row's data and newline,
// the file ends after the last row).  Only then is the result published; otherwise (short rows,
// headers of very different lengths, anything odd) k_index_rows walks the chain as before.
constexpr u64 IDX_NONE = ~0ull;
__device__ __forceinline__ u32 chunk_eq16(const uint4& a, uint32_t cccc);   // 16-bit mask: bytes equal to c
__global__ void __launch_bounds__(256) k_index_spec(const uint8_t* __restrict__ f, u64 n, const MsaHdr* h,
                                                    u64* __restrict__ hpos, u64* __restrict__ cand, u64 row_cap)
{
    if (h->status || h->first_hdr2 >= n) return;
    const u32 lane = threadIdx.x & 63;
    const u64 start0 = h->hdr_end + 1;
    const u64 Draw = h->first_hdr2 - start0;
    const u64 stride0 = h->first_hdr2 + 1;                     // header 0 -> header 1
    const u64 rmax = std::min<u64>(row_cap, n / (Draw + 3) + 2);
    const u64 wave = (blockIdx.x * (u64)blockDim.x + threadIdx.x) >> 6, nwaves = ((u64)gridDim.x * blockDim.x) >> 6;
    for (u64 r = 1 + wave; r < rmax; r += nwaves) {
        u64 found = IDX_NONE, st = IDX_NONE;
        const u64 g = r * stride0, rad = 64 + 8 * r;
        if (2 * rad + 128 < Draw && g < n + rad) {
            const u64 lo = g > rad ? g - rad : 1, hi = std::min<u64>(g + rad, n);
            u32 cnt = 0;
            for (u64 base = lo; base < hi; base += 1024) {         // 16 positions per lane and step
                const u64 i = base + (u64)lane * 16;
                u32 m = 0;
                if (i + 16 <= hi) {
                    const uint4 v = load16u(f + i);
                    const u32 gt = chunk_eq16(v, 0x3e3e3e3eu), nl = chunk_eq16(v, 0x0a0a0a0au);
                    m = gt & ((nl << 1) | (f[i - 1] == '\n' ? 1u : 0u)) & 0xffffu;
                } else {
                    for (u64 q = i; q < hi; q++) if (f[q] == '>' && f[q - 1] == '\n') m |= 1u << (q - i);
                }
                const u64 b = ballot64(m != 0);
                if (b) {
                    const int l0 = __builtin_ctzll(b);
                    const u32 m0 = (u32)__builtin_amdgcn_readlane((int)m, l0);
                    found = base + (u64)l0 * 16 + (u64)__builtin_ctz(m0);
                    u32 c = (u32)__builtin_popcount(m);
                    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
                    cnt += c;
                }
            }
            if (cnt != 1) found = cnt ? IDX_NONE - 1 : IDX_NONE;   // ambiguous / none
            else {
                for (u64 base = found; base < std::min<u64>(found + 4096, n); base += 64) {
                    const u64 i = base + lane;
                    const u64 b = ballot64(i < n && f[i] == '\n');
                    if (b) { st = base + (u64)__builtin_ctzll(b) + 1; break; }
                }
            }
        }
        if (lane == 0) { hpos[r] = found; cand[r] = st; }
    }
}
// one workgroup: link checks in parallel, then thread 0 decides
__global__ void __launch_bounds__(1024) k_index_check(const uint8_t* __restrict__ f, u64 n, MsaHdr* h,
                                                      const u64* __restrict__ hpos, const u64* __restrict__ cand,
                                                      u64* __restrict__ row_start, u64 row_cap)
{
    __shared__ u64 first_bad;
    if (h->status || h->first_hdr2 >= n) return;
    const u64 start0 = h->hdr_end + 1;
    const u64 Draw = h->first_hdr2 - start0;
    const u64 rmax = std::min<u64>(row_cap, n / (Draw + 3) + 2);
    if (threadIdx.x == 0) first_bad = rmax;
    __syncthreads();
    // row r is linked iff its header sits right behind row r-1's data + newline and has a data start
    for (u64 r = 1 + threadIdx.x; r < rmax; r += blockDim.x) {
        const u64 prev = r == 1 ? start0 : cand[r - 1];
        const bool ok = prev != IDX_NONE && hpos[r] == prev + Draw + 1 && cand[r] != IDX_NONE && cand[r] + Draw <= n;
        if (!ok) atomicMin(&first_bad, r);
    }
    __syncthreads();
    const u64 S = first_bad;                                   // rows 0 .. S-1 chain; row S must not exist
    if (S < 2 || S >= rmax) return;                            // (S >= rmax: could not see the end)
    if (hpos[S] != IDX_NONE) return;                           // something was found there but did not link
    const u64 q = (S == 1 ? start0 : cand[S - 1]) + Draw;      // behind the last row's data
    __shared__ u32 tail_bad;
    if (threadIdx.x == 0) tail_bad = 0;
    __syncthreads();
    if (q < n) {                                               // "\n" and then nothing but blank lines (<= 4096 bytes)
        if (n - q > 4097) { if (threadIdx.x == 0) tail_bad = 1; }
        else for (u64 i = q + threadIdx.x; i < n; i += blockDim.x) if (f[i] != '\n') tail_bad = 1;
    }
    __syncthreads();
    if (tail_bad) return;
    for (u64 r = threadIdx.x; r < S; r += blockDim.x) row_start[r] = r ? cand[r] : start0;
    if (threadIdx.x == 0) { h->idx_bad = S; h->S = S; __threadfence(); h->idx_done = 1; }
}

__global__ void k_index_rows(const uint8_t* __restrict__ f, u64 n, MsaHdr* h,
                             u64* __restrict__ row_start, u64 row_cap)
{
    const u32 lane = threadIdx.x;
    if (h->status) return;
    const u64 start0 = h->hdr_end + 1;
    if (h->first_hdr2 >= n) { if (lane == 0) h->status |= ST_FEW_ROWS; return; }
    const u64 lw = h->first_nl - start0;
    const u64 Draw = h->first_hdr2 - start0;        // raw bytes of one row, final newline excluded
    u64 L, wrapped;
    if (lw == 0) { if (lane == 0) h->status |= ST_LAYOUT; return; }
    if (Draw == lw) { L = lw; wrapped = 0; }
    else {
        u64 nlines = (Draw + 1 + lw) / (lw + 1);
        L = Draw + 1 - nlines;
        wrapped = 1;
        if (L == 0 || (L - 1) / lw != nlines - 1) { if (lane == 0) h->status |= ST_LAYOUT; return; }
    }
    // One dependent load per row: the 64-byte window at q = end of the previous row's data holds that
    // row's final newline, the next header's '>' and (headers are short) the header's newline.
    u64 s = 0, bad = 0;
    u64 st = start0;                                   // row 0: its header was found by k_find_hdr_end
    const bool spec = h->idx_done != 0;                // k_index_check validated the parallel index
    if (spec) s = h->S;
    while (!spec) {
        if (st + Draw > n) { bad = ST_LAYOUT; break; }
        if (s >= row_cap) { bad = ST_TOO_MANY_ROWS; break; }
        if (lane == 0) row_start[s] = st;
        s++;
        const u64 q = st + Draw;
        if (q == n) break;                             // no trailing newline
        const u32 c = q + lane < n ? f[q + lane] : 0x100u;
        if ((u32)__builtin_amdgcn_readlane((int)c, 0) != '\n') { bad = ST_LAYOUT; break; }
        const u64 p = q + 1;
        if (p == n) break;
        if ((u32)__builtin_amdgcn_readlane((int)c, 1) != '>') {
            // tolerate blank lines at the very end (skipped by the reference, :47-49)
            u64 rest = n - p;
            if (rest > 4096) { bad = ST_LAYOUT; break; }
            u64 nonnl = 0;
            for (u64 i = lane; i < rest; i += 64) nonnl |= (f[p + i] != '\n');
            if (ballot64(nonnl != 0)) bad = ST_LAYOUT;
            break;
        }
        u64 nlpos = n;
        u64 b = ballot64(lane >= 2 && c == '\n');
        if (b) nlpos = q + __builtin_ctzll(b);
        else {
            for (u64 base = q + 64; base < n; base += 64) {          // a header longer than the window
                const u64 i = base + lane;
                b = ballot64(i < n && f[i] == '\n');
                if (b) { nlpos = base + __builtin_ctzll(b); break; }
            }
        }
        if (nlpos >= n) { bad = ST_LAYOUT; break; }
        st = nlpos + 1;
    }
    if (lane == 0) {
        if (s < 2 && !bad) bad = ST_FEW_ROWS;
        h->status |= bad;
        h->S = s; h->L = L; h->lw = wrapped ? lw : 0; h->Draw = Draw;
        h->nwords = (L + 63) / 64;
        h->nwords_raw = (Draw + 63) / 64;
    }
}

// rows S .. n-1 of the row-start table repeat row S-1: the column scan's threads load 16 consecutive row starts
__global__ void k_pad_rows(u64* __restrict__ row_start, u64 S, u64 n)
{
    const u64 last = row_start[S - 1];
    for (u64 r = S + blockIdx.x * (u64)blockDim.x + threadIdx.x; r < n; r += (u64)gridDim.x * blockDim.x) row_start[r] = last;
}

// ---------------------------------------------------------------------------------------------
// K1: column scan + variant-column extraction.  One workgroup owns a tile of W = 16*CPR raw
// columns for ALL rows: T threads, thread (sub, j) holds the 16-byte chunk j of rows
// sub, sub+RI, ... (RI = T/CPR) in registers, so each input byte is read from HBM once.
//   msa_transforms.cpp:71-79  B[i] = 0 if c != ref[i] || c == '-'
// ---------------------------------------------------------------------------------------------
struct K1Params {
    const uint8_t* file; const u64* row_start; MsaHdr* hdr;
    u64* Vraw; u64* word_slot; uint8_t* vc; u64 vc_cap_cols;
    u64 Draw, lw; u32 S, Spad, cpr_log2, cap_cols /* LDS colbuf capacity in columns */;
    u64 ntiles;
    // fused grouping (context length 0, one-line rows, S <= 1024): the variant runs that lie inside a tile are
    // grouped right here, from the LDS image of the tile's variant columns; only the other columns go to vc
    u32 fuse; u64* Fraw; u32* rec_info; uint8_t* recf; u32 recf_stride, recf_gid;
};
constexpr u32 FUSE_MAXW = 10;      // widest run grouped by the column scan (exact 3-bit-per-column keys in one dword; two-dword keys
                                   // for 11..20 columns spill 71 registers here - rounds 2 and 3)
#ifndef EDSX_TAIL_WAVES
#define EDSX_TAIL_WAVES 8
#endif
constexpr u32 TAIL_WAVES = EDSX_TAIL_WAVES;   // waves of a scan workgroup that copy / group its variant columns (the rest retire early)
constexpr u32 CLIST = 2048;        // variant columns per tile in fused mode (the LDS image holds at most 64 KB / 32 B columns)
// fused record (indexed by the vc slot of the run's first column): group ids, 2 bits each (dword l = rows 16l..16l+15)
// for up to 4 strings, 4 bits each (two dwords per lane) for 5..16; then at recf_gid: u32 k | textlen << 8, then the
// .eds text "{s0,s1,..}" (<= REC_TEXT_MAX bytes).  rec_info[slot] = k | textlen << 8 | 4-bit ids << 30 | ok << 31
constexpr u32 REC_TEXT_MAX = 64;

__device__ __forceinline__ u32 chunk_ne16(const uint4& a, const uint4& b)
{
    return ne_bytes4(a.x, b.x) | (ne_bytes4(a.y, b.y) << 4) | (ne_bytes4(a.z, b.z) << 8) |
           (ne_bytes4(a.w, b.w) << 12);
}
__device__ __forceinline__ u32 chunk_eq16(const uint4& a, uint32_t cccc)
{
    return eq_byte4(a.x, cccc) | (eq_byte4(a.y, cccc) << 4) | (eq_byte4(a.z, cccc) << 8) |
           (eq_byte4(a.w, cccc) << 12);
}

// static byte extraction (a dynamic byte index makes hipcc keep the held chunks in scratch)
template <int I> __device__ __forceinline__ u32 byte_at(const uint4& v)
{
    const uint32_t w = I < 4 ? v.x : (I < 8 ? v.y : (I < 12 ? v.z : v.w));
    return (w >> ((I & 3) * 8)) & 0xffu;
}

// ---- grouping primitives of the wave-per-segment code (used by the column scan below for the segments it
// groups itself, and by k_seg_group): one wave per variant segment, lane l owns rows 16l .. 16l+15
constexpr int KCAP = 64;                  // distinct strings per fast segment (group g lives in lane g)
__device__ __forceinline__ uint32_t bytes_ne_mask(uint32_t a, uint32_t b)   // 0xFF where bytes differ
{
    uint32_t x = a ^ b;
    uint32_t h = (((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u;   // 0x80 where the bytes differ
    return h | (h - (h >> 7));              // -> 0xFF; (h >> 7) * 0xff would be a quarter-rate v_mul_lo_u32
}
__device__ __forceinline__ uint4 bytes_eq_mask(const uint4& a, uint32_t cccc)
{
    return make_uint4(~bytes_ne_mask(a.x, cccc), ~bytes_ne_mask(a.y, cccc), ~bytes_ne_mask(a.z, cccc),
                      ~bytes_ne_mask(a.w, cccc));
}
__device__ __forceinline__ bool any4(const uint4& v) { return (v.x | v.y | v.z | v.w) != 0; }
// does this lane hold a NUL byte in one of its existing rows?
__device__ __forceinline__ bool any_nul(const uint4& c, const uint4& vmask)
{
    const uint4 z = bytes_eq_mask(c, 0u);
    return ((z.x & vmask.x) | (z.y & vmask.y) | (z.z & vmask.z) | (z.w & vmask.w)) != 0;
}
__device__ __forceinline__ u32 first_byte_index(const uint4& m)   // m bytes are 0x00 / 0xFF
{
    return m.x ? (u32)__builtin_ctz(m.x) >> 3
               : m.y ? 4u + ((u32)__builtin_ctz(m.y) >> 3)
                     : m.z ? 8u + ((u32)__builtin_ctz(m.z) >> 3) : m.w ? 12u + ((u32)__builtin_ctz(m.w) >> 3) : 16u;
}
// byte `idx` (wave-uniform) of lane `leader`'s 16-byte vector, as a wave-uniform value
__device__ __forceinline__ u32 leader_byte(const uint4& v, int leader, u32 idx)
{
    const u32 x = (u32)__builtin_amdgcn_readlane((int)v.x, leader), y = (u32)__builtin_amdgcn_readlane((int)v.y, leader);
    const u32 z = (u32)__builtin_amdgcn_readlane((int)v.z, leader), w = (u32)__builtin_amdgcn_readlane((int)v.w, leader);
    const u32 d = idx < 8 ? (idx < 4 ? x : y) : (idx < 12 ? z : w);
    return (d >> ((idx & 3) * 8)) & 0xffu;
}
// '-' and '\n' contribute nothing to a row's string (msa_transforms.cpp:283): normalise to 0
template <bool CHECK_NL>
__device__ __forceinline__ uint4 normalise_col(const uint4& c, const uint4& vmask, u32& saw_nl)
{
    uint4 gap = bytes_eq_mask(c, 0x2d2d2d2du);
    if (CHECK_NL) {
        uint4 nl = bytes_eq_mask(c, 0x0a0a0a0au);
        if (any4(make_uint4(nl.x & vmask.x, nl.y & vmask.y, nl.z & vmask.z, nl.w & vmask.w))) saw_nl = 1;
        gap.x |= nl.x; gap.y |= nl.y; gap.z |= nl.z; gap.w |= nl.w;
    }
    return make_uint4(c.x & ~gap.x, c.y & ~gap.y, c.z & ~gap.z, c.w & ~gap.w);
}

struct FastGroups {
    uint4 gid;            // byte i = group of row 16*lane+i (0xFF: no such row)
    u32 k;                // number of distinct strings (wave-uniform)
    u32 sumlen;           // sum of their lengths
    // lane g holds the state of group g
    u64 key_lo, key_hi;   // the group's gap-stripped string (packed) or its hash, + length
    u32 rep;              // representative row (first row of the group in row order)
    u32 len;              // length of the group's string
};

// lane-private validity mask: byte i = 0xFF iff row 16*lane+i exists
__device__ __forceinline__ uint4 fast_valid_mask(u32 lane, u32 S)
{
    const u32 base = lane * 16u;
    const u32 n = S > base ? (S - base < 16u ? S - base : 16u) : 0u;       // existing rows of this lane
    auto word = [&](u32 o) -> uint32_t { return n >= o + 4u ? 0xffffffffu : (n > o ? (1u << (8u * (n - o))) - 1u : 0u); };
    return make_uint4(word(0), word(4), word(8), word(12));
}

// assign the rows in `eq` to the group with key (klo,khi): an existing one or a new one.
// Returns false when KCAP is exceeded.
__device__ __forceinline__ bool fast_assign(FastGroups& G, uint4& rm, const uint4& eq, u64 klo, u64 khi,
                                            u32 len, u32 lane, u32 rep_row)
{
    const u64 hit = ballot64(lane < G.k && G.key_lo == klo && G.key_hi == khi);
    u32 gsel;
    if (hit) gsel = (u32)__builtin_ctzll(hit);
    else {
        if (G.k >= (u32)KCAP) return false;
        gsel = G.k;
        if (lane == gsel) { G.key_lo = klo; G.key_hi = khi; G.rep = rep_row; G.len = len; }
        G.k++;
        G.sumlen += len;
    }
    const uint32_t gg = gsel * 0x01010101u;
    G.gid.x = (G.gid.x & ~eq.x) | (eq.x & gg); G.gid.y = (G.gid.y & ~eq.y) | (eq.y & gg);
    G.gid.z = (G.gid.z & ~eq.z) | (eq.z & gg); G.gid.w = (G.gid.w & ~eq.w) | (eq.w & gg);
    rm.x &= ~eq.x; rm.y &= ~eq.y; rm.z &= ~eq.z; rm.w &= ~eq.w;
    return true;
}

// signature weights of the multi-column grouping: W_j(c), 24 bit, odd.  A compile-time table (read
// with scalar loads) instead of two v_mul_lo_u32 per weight.
struct FastWeights {
    u32 v[64 * 3];
    constexpr FastWeights() : v{} {
        for (u32 c = 0; c < 64; c++)
            for (u32 j = 0; j < 3; j++) {
                u32 x = (c + 1u) * 0x9e3779b1u + (j + 1u) * 0x85ebca77u;
                x ^= x >> 15; x *= 0x2c1b3c6du; x ^= x >> 12;
                v[c * 3 + j] = (x | 1u) & 0xffffffu;
            }
    }
};
__device__ const FastWeights FAST_W{};

template <int CTRL, int ROWMASK> __device__ __forceinline__ u32 dpp_move0(u32 v)
{
    return (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWMASK, 0xf, false);   // lanes without a source get 0
}
// XOR / OR of v over the 64 lanes, wave-uniform (DPP row shifts and broadcasts; lane 63 ends up with all)
__device__ __forceinline__ u32 wave_xor_all(u32 v)
{
    v ^= dpp_move0<0x111, 0xf>(v);       // row_shr:1
    v ^= dpp_move0<0x112, 0xf>(v);       // row_shr:2
    v ^= dpp_move0<0x114, 0xf>(v);       // row_shr:4
    v ^= dpp_move0<0x118, 0xf>(v);       // row_shr:8   -> lane 15 of every row: the row's XOR
    v ^= dpp_move0<0x142, 0xa>(v);       // row_bcast:15 -> rows 1, 3
    v ^= dpp_move0<0x143, 0xc>(v);       // row_bcast:31 -> rows 2, 3
    return (u32)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ u32 wave_or_all(u32 v)
{
    v |= dpp_move0<0x111, 0xf>(v); v |= dpp_move0<0x112, 0xf>(v); v |= dpp_move0<0x114, 0xf>(v);
    v |= dpp_move0<0x118, 0xf>(v); v |= dpp_move0<0x142, 0xa>(v); v |= dpp_move0<0x143, 0xc>(v);
    return (u32)__builtin_amdgcn_readlane((int)v, 63);
}
// inclusive prefix sum over the 64 lanes
__device__ __forceinline__ u32 wave_scan_incl(u32 v)
{
    v += dpp_move0<0x111, 0xf>(v);       // row_shr:1
    v += dpp_move0<0x112, 0xf>(v);       // row_shr:2
    v += dpp_move0<0x114, 0xf>(v);       // row_shr:4
    v += dpp_move0<0x118, 0xf>(v);       // row_shr:8
    v += dpp_move0<0x142, 0xa>(v);       // row_bcast:15 -> rows 1, 3
    v += dpp_move0<0x143, 0xc>(v);       // row_bcast:31 -> rows 2, 3
    return v;
}
// four independent inclusive prefix sums, step by step side by side (a DPP instruction needs two wait states behind the
// write of its source: the other three scans fill them)
__device__ __forceinline__ void wave_scan_incl4(u32& a, u32& b, u32& c, u32& d)
{
#define EDSX_STEP(CTRL, RM) { const u32 ta = dpp_move0<CTRL, RM>(a), tb = dpp_move0<CTRL, RM>(b), tc = dpp_move0<CTRL, RM>(c), td = dpp_move0<CTRL, RM>(d); \
                              a += ta; b += tb; c += tc; d += td; }
    EDSX_STEP(0x111, 0xf) EDSX_STEP(0x112, 0xf) EDSX_STEP(0x114, 0xf) EDSX_STEP(0x118, 0xf) EDSX_STEP(0x142, 0xa) EDSX_STEP(0x143, 0xc)
#undef EDSX_STEP
}
// minimum of v over lanes 0..7 (wave-uniform)
__device__ __forceinline__ u32 min_lanes8(u32 v)
{
    u32 t;
    t = (u32)__builtin_amdgcn_update_dpp(-1, (int)v, 0x111, 0xf, 0xf, false); v = t < v ? t : v;
    t = (u32)__builtin_amdgcn_update_dpp(-1, (int)v, 0x112, 0xf, 0xf, false); v = t < v ? t : v;
    t = (u32)__builtin_amdgcn_update_dpp(-1, (int)v, 0x114, 0xf, 0xf, false); v = t < v ? t : v;
    return (u32)__builtin_amdgcn_readlane((int)v, 7);
}

// class code of the DNA alphabet {A, C, G, T, N, -}: class(b) = ((b >> 1) ^ (b >> 2)) & 7 :
//   A 0, C 1, G 2, N 4, '-' 5, T 7  (3 and 6 unused).  Any other byte (lower case, IUPAC codes, NUL, a stray
// newline) fails the reverse lookup (v_perm_b32 = four lookups in an 8-entry table per instruction).
constexpr u32 DNA_LET_LO = 0x00474341u, DNA_LET_HI = 0x54002d4eu;      // class -> letter (0: unused class)
__device__ __forceinline__ uint4 dna_classes(const uint4& x)
{
    return make_uint4(((x.x >> 1) ^ (x.x >> 2)) & 0x07070707u, ((x.y >> 1) ^ (x.y >> 2)) & 0x07070707u,
                      ((x.z >> 1) ^ (x.z >> 2)) & 0x07070707u, ((x.w >> 1) ^ (x.w >> 2)) & 0x07070707u);
}
__device__ __forceinline__ u32 dna_bad(const uint4& x, const uint4& cls, const uint4& vmask)
{
    return ((__builtin_amdgcn_perm(DNA_LET_HI, DNA_LET_LO, cls.x) ^ x.x) & vmask.x) |
           ((__builtin_amdgcn_perm(DNA_LET_HI, DNA_LET_LO, cls.y) ^ x.y) & vmask.y) |
           ((__builtin_amdgcn_perm(DNA_LET_HI, DNA_LET_LO, cls.z) ^ x.z) & vmask.z) |
           ((__builtin_amdgcn_perm(DNA_LET_HI, DNA_LET_LO, cls.w) ^ x.w) & vmask.w);
}

// One column over {A, C, G, T, N, -}: the grouping of msa_transforms.cpp:262-293 with byte-table lookups.
//   rb   = the column's byte of row `lane` (rows 0..63 one per lane: most classes first appear there, and then
//          their first row is one ballot away)
// Group ids are the ranks of the classes by first row (no loop over the groups: lane c ranks class c against the
// other seven with readlanes); group g's state lands in lane g as in fast_assign.
__device__ __forceinline__ bool fast_group_dna1(const uint4& x, u32 rb, const uint4& vmask, u32 lane, u32 S, FastGroups& G)
{
    constexpr u32 OH_LO = 0x08040201u, OH_HI = 0x80402010u;        // class -> 1 << class
    uint4 cls = dna_classes(x);
    if (ballot64(dna_bad(x, cls, vmask) != 0)) return false;
    // classes present in this lane's rows, and in the column
    u32 pl = (__builtin_amdgcn_perm(OH_HI, OH_LO, cls.x) & vmask.x) | (__builtin_amdgcn_perm(OH_HI, OH_LO, cls.y) & vmask.y) |
             (__builtin_amdgcn_perm(OH_HI, OH_LO, cls.z) & vmask.z) | (__builtin_amdgcn_perm(OH_HI, OH_LO, cls.w) & vmask.w);
    pl |= pl >> 16; pl |= pl >> 8; pl &= 0xffu;
    const u32 P = wave_or_all(pl);
    const u32 rc = lane < S ? (((rb >> 1) ^ (rb >> 2)) & 7u) : 8u;  // class of row `lane`
    // first row of every class: lane c keeps class c's (classes that are absent: ~0)
    u32 firstv = 0xffffffffu;
    for (u32 mm = P; mm; mm &= mm - 1) {
        const u32 c = (u32)__builtin_ctz(mm);
        const u64 b = ballot64(rc == c);
        u32 f;
        if (b) f = (u32)__builtin_ctzll(b);
        else {                                             // not among the first 64 rows
            const int L = __builtin_ctzll(ballot64(((pl >> c) & 1u) != 0));
            uint4 e = bytes_eq_mask(cls, c * 0x01010101u);
            e.x &= vmask.x; e.y &= vmask.y; e.z &= vmask.z; e.w &= vmask.w;
            f = 16u * (u32)L + (u32)__builtin_amdgcn_readlane((int)first_byte_index(e), L);
        }
        firstv = lane == c ? f : firstv;
    }
    // rank of class `lane` by first row = its group id (first rows are distinct)
    u32 rank = 0;
#pragma unroll
    for (int c = 0; c < 8; c++) rank += (u32)__builtin_amdgcn_readlane((int)firstv, c) < firstv ? 1u : 0u;
    // table class -> group: byte c of (lut_hi:lut_lo); OR over lanes 0..3 / 4..7 (row_shr within the first DPP row)
    u32 vlo = lane < 4u ? rank << (8u * lane) : 0u, vhi = (lane >= 4u && lane < 8u) ? rank << (8u * (lane - 4u)) : 0u;
    vlo |= dpp_move0<0x111, 0xf>(vlo); vhi |= dpp_move0<0x111, 0xf>(vhi);
    vlo |= dpp_move0<0x112, 0xf>(vlo); vhi |= dpp_move0<0x112, 0xf>(vhi);
    vlo |= dpp_move0<0x114, 0xf>(vlo); vhi |= dpp_move0<0x114, 0xf>(vhi);
    const u32 lut_lo = (u32)__builtin_amdgcn_readlane((int)vlo, 7), lut_hi = (u32)__builtin_amdgcn_readlane((int)vhi, 7);
    cls.x |= ~vmask.x; cls.y |= ~vmask.y; cls.z |= ~vmask.z; cls.w |= ~vmask.w;   // rows that do not exist: 0xFF
    G.gid = make_uint4(__builtin_amdgcn_perm(lut_hi, lut_lo, cls.x), __builtin_amdgcn_perm(lut_hi, lut_lo, cls.y),
                       __builtin_amdgcn_perm(lut_hi, lut_lo, cls.z), __builtin_amdgcn_perm(lut_hi, lut_lo, cls.w));
    // group g's letter, first row and length -> lane g
    G.key_lo = 0; G.key_hi = 0; G.rep = 0; G.len = 0;
    for (u32 mm = P; mm; mm &= mm - 1) {
        const u32 c = (u32)__builtin_ctz(mm);
        const u32 r = (u32)__builtin_amdgcn_readlane((int)rank, (int)c), f = (u32)__builtin_amdgcn_readlane((int)firstv, (int)c);
        const u32 letter = c == 5u ? 0u : (u32)((((u64)DNA_LET_HI << 32) | DNA_LET_LO) >> (8u * c)) & 0xffu;
        if (lane == r) { G.key_lo = letter; G.rep = f; G.len = letter ? 1u : 0u; }
    }
    G.k = (u32)__builtin_popcount(P);
    G.sumlen = (u32)__builtin_popcount(P & ~(1u << 5));
    return true;
}

// the not yet grouped row that comes first in row order (lane, byte): its lane and byte index (uniform)
__device__ __forceinline__ bool first_remaining(const uint4& rm, int& leader, u32& i0)
{
    const u64 b = ballot64(any4(rm));
    if (!b) return false;
    leader = __builtin_ctzll(b);
    i0 = (u32)__builtin_amdgcn_readlane((int)first_byte_index(rm), leader);
    return true;
}

// 2..20 columns over {A,C,G,T,N,-}: EXACT raw keys, 3 bits per column (class code), NK dwords of ten columns
// per row.  Rows with equal keys are byte-identical; a raw group's gap-stripped string is read off its key
// (drop the gap classes), so no row is re-read.  Returns 1 done, 0 another alphabet, -1 more than KCAP strings.
template <int NK, class LoadCol>
__device__ __forceinline__ int fast_group_dnakeys(LoadCol load_col, u32 ncol, const uint4& col0, u32 lane, const uint4& vmask,
                                                  FastGroups& G)
{
    u32 key[NK][16];
#pragma unroll
    for (int n = 0; n < NK; n++)
#pragma unroll
        for (int i = 0; i < 16; i++) key[n][i] = 0;
    u32 badacc = 0;
#define EDSX_K(I) key[n][I] |= byte_at<I>(cls) << sh;
#pragma unroll
    for (int n = 0; n < NK; n++) {
        const u32 cbase = 10u * n, cend = ncol < cbase + 10u ? ncol : cbase + 10u;
        for (u32 c0 = cbase; c0 < cend; c0 += 4) {
            uint4 cvs[4];                              // four column loads in flight
#pragma unroll
            for (int j = 0; j < 4; j++) {
                cvs[j] = make_uint4(0, 0, 0, 0);
                if (c0 + j < cend) cvs[j] = (c0 + j == 0) ? col0 : load_col(c0 + j);
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (c0 + j < cend) {
                    const uint4 x = cvs[j];
                    const uint4 cls = dna_classes(x);
                    badacc |= dna_bad(x, cls, vmask);
                    const u32 sh = 3u * (c0 + j - cbase);
                    EDSX_K(0) EDSX_K(1) EDSX_K(2) EDSX_K(3) EDSX_K(4) EDSX_K(5) EDSX_K(6) EDSX_K(7)
                    EDSX_K(8) EDSX_K(9) EDSX_K(10) EDSX_K(11) EDSX_K(12) EDSX_K(13) EDSX_K(14) EDSX_K(15)
                }
            }
        }
    }
#undef EDSX_K
    if (ballot64(badacc != 0)) return 0;
    uint4 rm = vmask;
    int leader;
    u32 i0;
    while (first_remaining(rm, leader, i0)) {
        u32 rk[NK];
#pragma unroll
        for (int n = 0; n < NK; n++) {
            u32 mk = 0;
#pragma unroll
            for (int i = 0; i < 16; i++) mk = (i0 == (u32)i) ? key[n][i] : mk;
            rk[n] = (u32)__builtin_amdgcn_readlane((int)mk, leader);
        }
        uint32_t e[4] = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 16; i++) {
            bool same = key[0][i] == rk[0];
            if (NK > 1) same = same && key[NK - 1][i] == rk[NK - 1];
            if (same) e[i >> 2] |= 0xffu << ((i & 3) * 8);
        }
        const uint4 eq = make_uint4(e[0] & rm.x, e[1] & rm.y, e[2] & rm.z, e[3] & rm.w);
        // the group's string: its key without the gap classes; lane = column
        u32 cl = 5u;
        if (lane < ncol) cl = ((lane < 10u ? rk[0] : rk[NK - 1]) >> (3u * (lane < 10u ? lane : lane - 10u))) & 7u;
        const u64 nz = ballot64(cl != 5u);
        const u32 len = (u32)__builtin_popcountll(nz), pos = mbcnt(nz);
        u32 klo = 0, khi = 0;                              // 3 bits per letter, ten letters per dword
        if (cl != 5u) { if (pos < 10u) klo = cl << (3u * pos); else khi = cl << (3u * (pos - 10u)); }
        klo = wave_or_all(klo);
        if (NK > 1) khi = wave_or_all(khi);
        if (!fast_assign(G, rm, eq, ((u64)khi << 32) | klo, (u64)len << 32, len, lane, (u32)leader * 16u + i0)) return -1;
    }
    return 1;
}


// group-id bytes of this lane's 16 rows -> 2 bits per row (ids 0..3; rows that do not exist: 0)
__device__ __forceinline__ u32 pack_gid2(const uint4& gid, const uint4& vmask)
{
    auto p = [](uint32_t x) -> u32 { x &= 0x03030303u; x |= x >> 6; x |= x >> 12; return x & 0xffu; };
    return p(gid.x & vmask.x) | (p(gid.y & vmask.y) << 8) | (p(gid.z & vmask.z) << 16) | (p(gid.w & vmask.w) << 24);
}
// ... -> 4 bits per row (ids 0..15): rows 0..7 in .x, 8..15 in .y
__device__ __forceinline__ uint2 pack_gid4(const uint4& gid, const uint4& vmask)
{
    auto p = [](uint32_t x) -> u32 { x &= 0x0f0f0f0fu; x |= x >> 4; return (x & 0xffu) | ((x >> 8) & 0xff00u); };
    return make_uint2(p(gid.x & vmask.x) | (p(gid.y & vmask.y) << 16), p(gid.z & vmask.z) | (p(gid.w & vmask.w) << 16));
}

// One run of variant columns that lies inside a tile of the column scan, grouped by one wave from the LDS image of
// the tile's variant columns (colbuf, column-major, natural row order): lane l = rows 16l .. 16l+15
// (msa_transforms.cpp:262-293).  desc = index of the run's first column in colbuf | width << 11.  Writes the fused
// record (group ids + .eds text) and rec_info, or - when the run is not for this path (another alphabet, more than
// 16 strings, a long text) - copies its columns to vc for the grouping kernels.
// ROWS64 (S <= 64): one row per lane instead of sixteen - a column is one byte per lane, the distinct strings fall out
// of a ballot per string, and runs of up to 20 columns are taken (exact 3-bit keys of the gap-stripped strings, ten
// letters per dword).
template <bool ROWS64>
__device__ __forceinline__ void fused_group_run(const K1Params& p, const uint8_t* colbuf, u32 desc, u64 slot_base, u32 lane,
                                                const uint4& vmask, u32 nl, u32 loff)
{
    const u32 idx0 = desc & 0x7ffu, w = desc >> 11;
    const uint8_t* c0p = colbuf + (size_t)idx0 * p.Spad;
    FastGroups G;
    G.gid = make_uint4(~0u, ~0u, ~0u, ~0u);
    G.k = 0; G.sumlen = 0; G.key_lo = 0; G.key_hi = 0; G.rep = 0; G.len = 0;
    bool ok;
    if constexpr (ROWS64) {
        const bool act = lane < p.S;
        const u64 LET = ((u64)DNA_LET_HI << 32) | DNA_LET_LO;
        u32 klo = 0, khi = 0, len = 0, bad = 0;
        for (u32 c = 0; c < w; c++) {                          // the gap-stripped string of row `lane` as class codes
            const u32 ch = act ? (u32)c0p[(size_t)c * p.Spad + lane] : (u32)'-';
            const u32 cls = ((ch >> 1) ^ (ch >> 2)) & 7u;
            bad |= ((u32)(LET >> (8u * cls)) & 0xffu) != ch ? 1u : 0u;
            if (cls != 5u) { if (len < 10u) klo |= cls << (3u * len); else khi |= cls << (3u * (len - 10u)); len++; }
        }
        ok = !ballot64(act && bad);
        u32 mygid = 0, gk_lo = 0, gk_hi = 0, glen = 0;         // lane g: string g
        if (ok) {
            u64 todo = ballot64(act);
            while (todo) {                                     // strings in the order of their first rows (msa_transforms.cpp:288-293)
                const int leader = __builtin_ctzll(todo);
                const u32 a0 = (u32)__builtin_amdgcn_readlane((int)klo, leader), a1 = (u32)__builtin_amdgcn_readlane((int)khi, leader);
                const u32 l0 = (u32)__builtin_amdgcn_readlane((int)len, leader);
                const u64 m = ballot64(act && klo == a0 && khi == a1 && len == l0);
                if ((m >> lane) & 1ull) mygid = G.k;
                if (lane == G.k) { gk_lo = a0; gk_hi = a1; glen = l0; }
                G.k++; G.sumlen += l0;
                todo &= ~m;
            }
            G.key_lo = ((u64)gk_hi << 32) | gk_lo; G.len = glen;
            // group-id bytes of rows 16l .. 16l+15 for the lanes that own a dword of the record
            uint32_t gb[4] = {0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const u32 v = (u32)__shfl((int)mygid, (int)((lane * 16u + (u32)i) & 63u), 64);
                gb[i >> 2] |= (v & 0xffu) << ((i & 3) * 8);
            }
            G.gid = make_uint4(gb[0], gb[1], gb[2], gb[3]);
        }
    } else {
    const uint4 col0 = *reinterpret_cast<const uint4*>(c0p + loff);
    if (w == 1u) ok = fast_group_dna1(col0, lane < p.S ? (u32)c0p[lane] : 0u, vmask, lane, p.S, G);
    else {
        auto load_col = [&](u32 c) -> uint4 { return *reinterpret_cast<const uint4*>(c0p + (size_t)c * p.Spad + loff); };
        ok = fast_group_dnakeys<1>(load_col, w, col0, lane, vmask, G) > 0;
    }
    }
    const u32 textlen = 1u + G.k + G.sumlen;             // "{" + strings + separators / "}"
    ok = ok && G.k <= 16u && textlen <= REC_TEXT_MAX;
    const u64 slot = slot_base + idx0;
    if (ok) {
        uint8_t* rec = p.recf + slot * (u64)p.recf_stride;
        if (lane < nl) {
            if (G.k <= 4u) *reinterpret_cast<u32*>(rec + lane * 4u) = pack_gid2(G.gid, vmask);
            else *reinterpret_cast<uint2*>(rec + lane * 8u) = pack_gid4(G.gid, vmask);
        }
        uint8_t* t = rec + p.recf_gid + 4;
        if (w == 1u && !ROWS64) {                  // lane g holds string g's letter (0: the empty string)
            const u32 c = lane < G.k ? (u32)G.key_lo : 0u;
            const u64 nz = ballot64(c != 0);
            const u32 at = 1u + lane + mbcnt(nz);
            if (lane < G.k) {
                if (c) t[at] = (uint8_t)c;
                t[at + (c ? 1u : 0u)] = lane + 1u < G.k ? ',' : '}';
            }
        } else {                                   // lane g holds string g as 3-bit classes
            const u32 mine = lane < G.k ? G.len + 1u : 0u;
            const u32 at = 1u + wave_scan_incl(mine) - mine;
            const u32 sk = (u32)G.key_lo, sk2 = (u32)(G.key_lo >> 32);     // ten letters per dword
            if (lane < G.k) {
                for (u32 i = 0; i < G.len; i++)
                    t[at + i] = (uint8_t)__builtin_amdgcn_perm(DNA_LET_HI, DNA_LET_LO,
                                                               i < 10u ? (sk >> (3u * i)) & 7u : (sk2 >> (3u * (i - 10u))) & 7u);
                t[at + G.len] = lane + 1u < G.k ? ',' : '}';
            }
        }
        if (lane == 0) {
            t[0] = '{';
            *reinterpret_cast<u32*>(rec + p.recf_gid) = G.k | (textlen << 8);
            p.rec_info[slot] = G.k | (textlen << 8) | (G.k > 4u ? 1u << 30 : 0u) | (1u << 31);
        }
    } else {                                       // not for this path: its columns go to vc after all
        if (lane == 0) p.rec_info[slot] = 0;
        for (u32 c = 0; c < w; c++)
            for (u32 o = lane * 16u; o < p.Spad; o += 1024u)
                *reinterpret_cast<uint4*>(p.vc + (slot + c) * (u64)p.Spad + o) =
                    *reinterpret_cast<const uint4*>(c0p + (size_t)c * p.Spad + o);
    }
}

// BIG: more rows than the LDS holds (no row-start table, no column image there): row starts are read from HBM, the
// variant bytes go straight to vc.
template <int T, int RPT, bool HOLD, bool LANEROWS, int MINW, bool ROWS64 = false, bool BIG = false>
__global__ void __launch_bounds__(T, MINW) k_scan_extract(K1Params p)
{
    static_assert(!BIG || (!HOLD && !LANEROWS && !ROWS64), "BIG is the plain variant");
    extern __shared__ __attribute__((aligned(16))) uint8_t colbuf[];
    __shared__ __attribute__((aligned(16))) u32 D[256];
    __shared__ u32 pre[256];
    __shared__ u32 wtot[4];
    __shared__ u64 slot_base_sh;
    __shared__ u32 CS[256];            // fused: per chunk, first columns of the runs grouped here
    __shared__ uint16_t clist[CLIST];  // fused: from the front those runs (colbuf index | width << 12), from the back the colbuf
    __shared__ u32 ncand_sh, nst_sh;   //        indices of the variant columns that go to vc; their numbers

    const u32 tid = threadIdx.x;
    const u32 cpr = 1u << p.cpr_log2;
    const u32 j = tid & (cpr - 1);
    const u32 sub = tid >> p.cpr_log2;
    const u32 RI = T >> p.cpr_log2;
    // XCD-aware tile order: workgroups b, b+8, b+16.. share an XCD (and its L2); give them
    // neighbouring tiles so the cache lines split by a tile edge are fetched from HBM once.
    u64 tile;
    {
        const u64 nt = p.ntiles, b = blockIdx.x;
        // (interleaving the XCDs' tile ranges instead, or plain blockIdx order, measured the same)
        const u64 per = nt / 8, rem = nt % 8;     // XCD x owns per (+1 if x < rem) tiles
        const u64 x = b % 8, k = b / 8;
        tile = x * per + (x < rem ? x : rem) + k;
    }
    const u64 q0 = tile * (u64)(cpr * 16);
    const u64 q = q0 + (u64)j * 16;
    const bool full_tile = q0 + (u64)cpr * 16 <= p.Draw;       // workgroup-uniform
    const int nb = q < p.Draw ? ((p.Draw - q) < 16 ? (int)(p.Draw - q) : 16) : 0;
    const u32 valid = nb == 16 ? 0xffffu : ((1u << nb) - 1u);

    // row starts -> LDS (the colbuf area is free until the extraction phase), so the data loads
    // below depend on fast ds_reads only and all RPT of them are in flight together
    // (Full tiles of the lane-rows layout take their 16 consecutive row starts straight from the padded table -
    // eight 16-byte loads that hit L1/L2 - so a wave issues its data loads without waiting for the workgroup.)
    constexpr bool DIRECT_OK = HOLD && LANEROWS;
    const bool direct = DIRECT_OK && full_tile;                // workgroup-uniform
    const u64* rs = BIG ? p.row_start : reinterpret_cast<const u64*>(colbuf);
    if (!direct && !BIG) for (u32 r = tid; r < p.S; r += T) reinterpret_cast<u64*>(colbuf)[r] = p.row_start[r];
    if (tid < 256) { D[tid] = 0; CS[tid] = 0; }
    if (tid == 0) { ncand_sh = 0; nst_sh = 0; }
    if (!direct) __syncthreads();

    const uint8_t* f = p.file;
    const u32 Sm1 = p.S - 1;
    // rows of this thread.  Plain: sub, sub + RI, ...  Lane rows (16 * RI >= S): 16 consecutive rows
    // 16*sub .. 16*sub+15, i.e. 16 consecutive bytes of a vc column (natural row order).
    auto row_of = [&](u32 it) -> u32 {
        if constexpr (LANEROWS) return sub * 16u + it;
        else return sub + it * RI;
    };
    uint4 ref = make_uint4(0, 0, 0, 0);
    uint4 d[HOLD ? RPT : 1];
    uint4 acc = make_uint4(0, 0, 0, 0);                        // OR over rows of (row ^ ref): a byte is
    // (Rows of a FASTA image sit at odd byte offsets, and 16-byte lane loads from addresses that are not multiples of 4
    // stream a quarter slower than dword-aligned ones - profiles/exp/scan_skel4.  Loading from the address rounded down
    // to a multiple of 4 and shifting the bytes into place with v_alignbyte_b32 + DPP was built and measured in round 3:
    // the ~180 extra instructions per thread sit exactly where the workgroup is issue-bound, 30.9 instead of 26.8 ms.)
    if (direct) {
        if constexpr (DIRECT_OK) {
            ref = load16u(f + p.row_start[0] + q);
            const ulonglong2* rp = reinterpret_cast<const ulonglong2*>(p.row_start + sub * 16u);   // padded: rows past S = row S-1
            ulonglong2 rv[8];
#pragma unroll
            for (int i = 0; i < 8; i++) rv[i] = rp[i];
#pragma unroll
            for (int i = 0; i < 8; i++) { d[2 * i] = load16u(f + rv[i].x + q); d[2 * i + 1] = load16u(f + rv[i].y + q); }
#pragma unroll
            for (int it = 0; it < RPT; it++) {
                acc.x |= d[it].x ^ ref.x; acc.y |= d[it].y ^ ref.y;
                acc.z |= d[it].z ^ ref.z; acc.w |= d[it].w ^ ref.w;
            }
        }
    } else if (full_tile) {                                    // non-zero iff some row differs there
        ref = load16u(f + rs[0] + q);                          // fast path: unconditional 16-B loads
#pragma unroll
        for (int it = 0; it < RPT; it++) {
            const u32 r = row_of(it);
            d[HOLD ? it : 0] = load16u(f + rs[r < p.S ? r : Sm1] + q);   // clamped: rows past S re-read row S-1
            if constexpr (!HOLD) {
                acc.x |= d[0].x ^ ref.x; acc.y |= d[0].y ^ ref.y; acc.z |= d[0].z ^ ref.z; acc.w |= d[0].w ^ ref.w;
            }
        }
        if constexpr (HOLD) {
#pragma unroll
            for (int it = 0; it < RPT; it++) {                 // a clamped duplicate changes nothing
                acc.x |= d[it].x ^ ref.x; acc.y |= d[it].y ^ ref.y;
                acc.z |= d[it].z ^ ref.z; acc.w |= d[it].w ^ ref.w;
            }
        }
    } else {
        if (nb > 0) ref = load_partial(f + rs[0] + q, nb);
#pragma unroll
        for (int it = 0; it < RPT; it++) {
            const u32 r = row_of(it);
            uint4 v = ref;
            if (r < p.S && nb > 0) v = load_partial(f + rs[r] + q, nb);
            d[HOLD ? it : 0] = v;
            acc.x |= v.x ^ ref.x; acc.y |= v.y ^ ref.y; acc.z |= v.z ^ ref.z; acc.w |= v.w ^ ref.w;
        }
    }
    if constexpr (!HOLD) {                                     // S beyond the register budget
        for (u32 r = sub + RPT * RI; r < p.S; r += RI) {
            if (nb > 0) {
                const uint8_t* src = f + rs[r] + q;
                uint4 v = nb == 16 ? load16u(src) : load_partial(src, nb);
                acc.x |= v.x ^ ref.x; acc.y |= v.y ^ ref.y; acc.z |= v.z ^ ref.z; acc.w |= v.w ^ ref.w;
            }
        }
    }
    u32 diff = chunk_ne16(acc, make_uint4(0, 0, 0, 0));
    diff |= chunk_eq16(ref, 0x2d2d2d2du);                      // '-' in row 0 => variant column
    diff &= valid;
    u32 bad = 0;
    u32 nlmask = 0;
    if (p.lw && nb > 0) {                                      // wrapped rows: newline positions
        u64 m = q % (p.lw + 1);
        for (int i = 0; i < nb; i++) { if (m == p.lw) { nlmask |= 1u << i; m = 0; } else m++; }
    }
    if ((chunk_eq16(ref, 0x0a0a0a0au) & valid) != nlmask) bad = 1;
    if (direct) {
        // OR over the lanes of the wave that hold the same chunk (lanes j, j + cpr, ..): one LDS atomic per wave and chunk
        for (u32 o = cpr; o < 64u; o <<= 1) diff |= (u32)__shfl_xor((int)diff, (int)o, 64);
        __syncthreads();                                       // D[] is zeroed (no barrier in front of the loads)
        if ((tid & 63u) < cpr && diff) atomicOr(&D[j], diff);
    } else if (diff) atomicOr(&D[j], diff);
    __syncthreads();

    const u32 V16 = D[j];
    if (V16 & nlmask) bad = 1;                                 // a row deviates at a newline slot

    // exclusive prefix of popc(D[*]) over the tile's chunks
    const bool fastpre = cpr <= 8u;                            // every thread derives it from the <= 8 masks itself
    u64 prepk = 0;                                             // byte c: variant columns in chunks 0 .. c-1
    u32 w0 = 0, w1 = 0, w2 = 0, w3 = 0, nv = 0;
    if (fastpre) {
        // (the masks are the same in every lane: as scalars their popcounts and the packing run on the scalar unit,
        // beside the vector work of the SIMD's other waves)
        const uint4 da = *reinterpret_cast<const uint4*>(&D[0]), db = *reinterpret_cast<const uint4*>(&D[4]);
        const u32 dm[8] = {uniform32(da.x), uniform32(da.y), uniform32(da.z), uniform32(da.w),
                           uniform32(db.x), uniform32(db.y), uniform32(db.z), uniform32(db.w)};
#pragma unroll
        for (int c = 0; c < 8; c++) { prepk |= (u64)nv << (8 * c); nv += (u32)__builtin_popcount(dm[c]); }
    } else {
        if (tid < 256) {                                       // (cpr <= 256 -> <= 4 waves)
            u32 c = tid < cpr ? __builtin_popcount(D[tid]) : 0;
            u32 incl = c;
            for (int o = 1; o < 64; o <<= 1) { u32 a = __shfl_up(incl, o, 64); if ((tid & 63) >= (u32)o) incl += a; }
            if ((tid & 63) == 63) wtot[tid >> 6] = incl;
            pre[tid] = incl - c;
        }
        __syncthreads();
        w0 = wtot[0]; w1 = wtot[1]; w2 = wtot[2]; w3 = wtot[3];
        nv = w0 + w1 + w2 + w3;
    }
    // (the slot atomic is issued now and its result is first needed after the extraction into LDS, which hides
    // its ~1 us round trip)
    auto pre_of = [&](u32 chunk) -> u32 {
        if (fastpre) return (u32)(prepk >> (8u * (chunk & 7u))) & 0xffu;
        const u32 cw = chunk >> 6;
        return pre[chunk] + (cw > 0 ? w0 : 0u) + (cw > 1 ? w1 : 0u) + (cw > 2 ? w2 : 0u);
    };
    u64 base_r = 0;
    if (tid == 0 && nv) base_r = atomicAdd(&p.hdr->nv, (u64)nv);

    // extraction: variant bytes -> LDS (column-major) -> HBM, in batches of cap_cols columns
    u64 slot_base = 0;
    bool overflow = false;
    {
        const u32 cap = p.cap_cols;
        // LANEROWS: this thread's 16 rows x 16 columns are transposed in registers with v_perm_b32 (two rounds of byte
        // interleaves per 4x4 block, 128 instructions), one dword component = four columns at a time: VISIT(I, a, b, c, d)
        // gets column I as four dwords (rows 0..3, 4..7, 8..11, 12..15 of the thread).  The held chunks and a whole
        // transposed copy are never live together.
#define EDSX_TCOMP(C, COMP, VISIT) {                                                               \
            uint32_t t4[4][4];                                                                     \
            _Pragma("unroll") for (int k4 = 0; k4 < 4; k4++) {                                    \
                uint32_t a0 = d[4 * k4].COMP, a1 = d[4 * k4 + 1].COMP, a2 = d[4 * k4 + 2].COMP, a3 = d[4 * k4 + 3].COMP; \
                /* opaque: or the optimiser hoists the permutes of all four components (they are the same in the fused */ \
                /* and the batched path) in front of the branch, and the transposed copy is live beside the chunks again */ \
                asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));                        \
                const uint32_t t0 = __builtin_amdgcn_perm(a1, a0, 0x05010400u), t1 = __builtin_amdgcn_perm(a1, a0, 0x07030602u); \
                const uint32_t t2 = __builtin_amdgcn_perm(a3, a2, 0x05010400u), t3 = __builtin_amdgcn_perm(a3, a2, 0x07030602u); \
                t4[0][k4] = __builtin_amdgcn_perm(t2, t0, 0x05040100u);                           \
                t4[1][k4] = __builtin_amdgcn_perm(t2, t0, 0x07060302u);                           \
                t4[2][k4] = __builtin_amdgcn_perm(t3, t1, 0x05040100u);                           \
                t4[3][k4] = __builtin_amdgcn_perm(t3, t1, 0x07060302u);                           \
            }                                                                                      \
            VISIT(4 * C, t4[0][0], t4[0][1], t4[0][2], t4[0][3]) VISIT(4 * C + 1, t4[1][0], t4[1][1], t4[1][2], t4[1][3]) \
            VISIT(4 * C + 2, t4[2][0], t4[2][1], t4[2][2], t4[2][3]) VISIT(4 * C + 3, t4[3][0], t4[3][1], t4[3][2], t4[3][3]) }
#define EDSX_TALL(VISIT) EDSX_TCOMP(0, x, VISIT) EDSX_TCOMP(1, y, VISIT) EDSX_TCOMP(2, z, VISIT) EDSX_TCOMP(3, w, VISIT)
        bool fused_tile = false;
        if constexpr (HOLD && LANEROWS) fused_tile = p.fuse && nv && nv <= cap && nv <= CLIST;   // workgroup-uniform
        if (fused_tile) { if constexpr (HOLD && LANEROWS) {
            // ---- all variant columns of the tile -> LDS (column-major, natural row order)
            if (V16) {
                // (threads whose 16 rows do not exist write into the 16 bytes of slack behind the column's rows)
                uint8_t* dst = colbuf + (size_t)pre_of(j) * p.Spad + (sub * 16u < p.Spad - 16u ? sub * 16u : p.Spad - 16u);
#define EDSX_VISIT(I, A, B, C_, D_) if (V16 & (1u << (I))) { *reinterpret_cast<uint4*>(dst) = make_uint4(A, B, C_, D_); dst += p.Spad; }
                EDSX_TALL(EDSX_VISIT)
#undef EDSX_VISIT
            }
            // ---- the runs of variant columns that lie inside the tile and are at most FUSE_MAXW wide are grouped here
            // (registered by the thread that owns their first column); every other variant column goes to vc
            for (u32 col = tid; col < cpr * 16u; col += T) {   // one thread per column of the tile
                const u32 ch = col >> 4, b = col & 15u;
                const u32 m = D[ch];
                if ((m >> b) & 1u) {
                    // window: previous, own and the next two chunks.  Beyond the tile the runs may go on: all ones there
                    const u64 w = (ch ? (u64)D[ch - 1] : 0xffffull) | ((u64)m << 16) | ((u64)(ch + 1 < cpr ? D[ch + 1] : 0xffffu) << 32) |
                                  ((u64)(ch + 2 < cpr ? D[ch + 2] : 0xffffu) << 48);
                    const u32 pp = 16u + b;
                    const u32 up = (u32)__builtin_ctzll(~(w >> pp));                 // ones from this column upwards
                    const u32 dn = (u32)__builtin_clzll(~(w << (64u - pp)));         // ones below it
                    const u32 idx = pre_of(ch) + (u32)__builtin_popcount(m & ((1u << b) - 1u));
                    // (a run that reaches the bottom of the window may be longer than it looks: not for this path.
                    // Upwards the window shows at least 32 columns, so the thread of a run's first column sees it whole.)
                    if (up + dn > (ROWS64 ? 20u : FUSE_MAXW) || dn == pp) clist[CLIST - 1u - atomicAdd(&nst_sh, 1u)] = (uint16_t)idx;
                    else if (dn == 0) { clist[atomicAdd(&ncand_sh, 1u)] = (uint16_t)(idx | ((up + dn) << 11)); atomicOr(&CS[ch], 1u << b); }
                }
            }
            if (tid == 0) {                                    // first use of the atomic's result
                slot_base_sh = base_r;
                if (base_r + nv > p.vc_cap_cols) atomicOr(&p.hdr->status, (u64)ST_VC_OVERFLOW);
            }
            __syncthreads();
            slot_base = slot_base_sh;
            overflow = slot_base + nv > p.vc_cap_cols;
            if (tid < cpr / 4) {                               // V words, per-word slot base, first columns of the fused runs
                u64 wi = q0 / 64 + tid;
                if (wi * 64 < p.Draw) {
                    p.Vraw[wi] = (u64)D[4 * tid] | ((u64)D[4 * tid + 1] << 16) | ((u64)D[4 * tid + 2] << 32) | ((u64)D[4 * tid + 3] << 48);
                    p.word_slot[wi] = slot_base + pre_of(4 * tid);
                    p.Fraw[wi] = (u64)CS[4 * tid] | ((u64)CS[4 * tid + 1] << 16) | ((u64)CS[4 * tid + 2] << 32) | ((u64)CS[4 * tid + 3] << 48);
                }
            }
            if (!overflow) {
                // ---- the other variant columns: LDS -> vc, one wave per column
                // (Only the first TAIL_WAVES waves - one or two per SIMD - do this tail: the others end here, and a
                // workgroup that is waiting for registers can start loading while these finish.)
                const u32 wv = uniform32(tid >> 6);
                if (wv >= TAIL_WAVES) {
                    if (bad) atomicOr(&p.hdr->status, (u64)(ST_LAYOUT | ST_NEWLINE_IN_DATA));
                    return;
                }
                {
                    const u32 nst = nst_sh, vec = p.Spad / 16u, ln = tid & 63u;
                    for (u32 c = wv; c < nst; c += TAIL_WAVES) {
                        const u32 idx = clist[CLIST - 1u - c];
                        const uint8_t* src = colbuf + (size_t)idx * p.Spad;
                        uint8_t* g = p.vc + (slot_base + idx) * (u64)p.Spad;
                        for (u32 pc = ln; pc < vec; pc += 64u)
                            *reinterpret_cast<uint4*>(g + pc * 16u) = *reinterpret_cast<const uint4*>(src + pc * 16u);
                    }
                }
                // ---- group the runs: one wave per run, lane l = rows 16l .. 16l+15 (msa_transforms.cpp:262-293)
                const u32 ncand = ncand_sh;
                const u32 lane = tid & 63u, nl = (p.S + 15u) >> 4;
                const uint4 vmask = fast_valid_mask(lane, p.S);
                const u32 loff = lane * 16u < p.Spad - 16u ? lane * 16u : p.Spad - 16u;
                for (u32 ci = wv; ci < ncand; ci += TAIL_WAVES)
                    fused_group_run<ROWS64>(p, colbuf, uniform32((u32)clist[ci]), slot_base, lane, vmask, nl, loff);
            }
        } } else if constexpr (BIG) {
            if (tid == 0) {
                slot_base_sh = base_r;
                if (base_r + nv > p.vc_cap_cols) atomicOr(&p.hdr->status, (u64)ST_VC_OVERFLOW);
            }
            __syncthreads();
            slot_base = slot_base_sh;
            overflow = slot_base + nv > p.vc_cap_cols;
            if (tid < (cpr + 3u) / 4u) {                       // V words + per-word slot base
                const u64 wi = q0 / 64 + tid;
                if (wi * 64 < p.Draw) {
                    p.Vraw[wi] = (u64)D[4 * tid] | ((u64)D[4 * tid + 1] << 16) | ((u64)D[4 * tid + 2] << 32) | ((u64)D[4 * tid + 3] << 48);
                    p.word_slot[wi] = slot_base + pre_of(4 * tid);
                }
            }
            if (nv && !overflow && V16) {
                u32 m = V16, idx = pre_of(j);
                while (m) {
                    const int i = __builtin_ctz(m);
                    m &= m - 1;
                    uint8_t* dst = p.vc + (slot_base + idx) * (u64)p.Spad;
                    for (u32 r = sub; r < p.S; r += RI) {
                        const u32 ch = f[p.row_start[r] + q + i];
                        if (ch == '\n') bad = 1;
                        dst[r] = (uint8_t)ch;
                    }
                    idx++;
                }
            }
        } else
        for (u32 b0 = 0; b0 < nv || b0 == 0; b0 += cap) {
            if (V16 && nv) {
                u32 idx = pre_of(j);
                if constexpr (HOLD && LANEROWS) {              // this thread's 16 rows are 16 consecutive bytes of the column
#define EDSX_VISIT(I, A, B, C_, D_)                                                             \
                    if (V16 & (1u << (I))) {                                                   \
                        if (idx >= b0 && idx < b0 + cap && sub * 16u < p.S)      /* (rows past S: slack) */ \
                            *reinterpret_cast<uint4*>(colbuf + (size_t)(idx - b0) * p.Spad + sub * 16) = make_uint4(A, B, C_, D_); \
                        idx++;                                                                 \
                    }
                    EDSX_TALL(EDSX_VISIT)
#undef EDSX_VISIT
                } else if constexpr (HOLD) {
#define EDSX_X(I)                                                                              \
                    if (V16 & (1u << I)) {                                                     \
                        if (idx >= b0 && idx < b0 + cap) {                                     \
                            uint8_t* dst = colbuf + (size_t)(idx - b0) * p.Spad;               \
                            _Pragma("unroll") for (int it = 0; it < RPT; it++) {               \
                                const u32 r = sub + it * RI;                                   \
                                if (r < p.S) {                                                 \
                                    const u32 ch = byte_at<I>(d[it]);                          \
                                    if (ch == '\n') bad = 1;                                   \
                                    dst[r] = (uint8_t)ch;                                      \
                                }                                                              \
                            }                                                                  \
                        }                                                                      \
                        idx++;                                                                 \
                    }
                    EDSX_X(0) EDSX_X(1) EDSX_X(2) EDSX_X(3) EDSX_X(4) EDSX_X(5) EDSX_X(6) EDSX_X(7)
                    EDSX_X(8) EDSX_X(9) EDSX_X(10) EDSX_X(11) EDSX_X(12) EDSX_X(13) EDSX_X(14) EDSX_X(15)
#undef EDSX_X
                } else {
                    u32 m = V16;
                    while (m) {
                        const int i = __builtin_ctz(m);
                        m &= m - 1;
                        if (idx >= b0 && idx < b0 + cap) {
                            uint8_t* dst = colbuf + (size_t)(idx - b0) * p.Spad;
                            for (u32 r = sub; r < p.S; r += RI) {
                                u32 ch = f[p.row_start[r] + q + i];
                                if (ch == '\n') bad = 1;
                                dst[r] = (uint8_t)ch;
                            }
                        }
                        idx++;
                    }
                }
            }
            if (b0 == 0 && tid == 0) {                         // first use of the atomic's result
                slot_base_sh = base_r;
                if (base_r + nv > p.vc_cap_cols) atomicOr(&p.hdr->status, (u64)ST_VC_OVERFLOW);
            }
            __syncthreads();
            if (b0 == 0) {
                slot_base = slot_base_sh;
                overflow = slot_base + nv > p.vc_cap_cols;
                if (tid < cpr / 4) {                           // V words + per-word slot base
                    u64 wi = q0 / 64 + tid;
                    if (wi * 64 < p.Draw) {
                        u64 bits = (u64)D[4 * tid] | ((u64)D[4 * tid + 1] << 16) | ((u64)D[4 * tid + 2] << 32) |
                                   ((u64)D[4 * tid + 3] << 48);
                        p.Vraw[wi] = bits;
                        p.word_slot[wi] = slot_base + pre_of(4 * tid);
                        if (p.fuse) p.Fraw[wi] = 0;
                    }
                }
            }
            if (!nv || overflow) break;                        // workgroup-uniform
            const u32 ncols = (nv - b0) < cap ? (nv - b0) : cap;
            const size_t nbytes = (size_t)ncols * p.Spad;      // Spad % 16 == 0
            uint8_t* g = p.vc + (slot_base + b0) * (u64)p.Spad;
            for (size_t o = (size_t)tid * 16; o < nbytes; o += (size_t)T * 16)
                *reinterpret_cast<uint4*>(g + o) = *reinterpret_cast<const uint4*>(colbuf + o);
            if (b0 + cap < nv) __syncthreads();                // colbuf is reused by the next batch
        }
    }
    if (bad) atomicOr(&p.hdr->status, (u64)(ST_LAYOUT | ST_NEWLINE_IN_DATA));
}

// alignment-space V from raw-space Vraw (wrapped rows only): drop the newline positions
__global__ void k_vmap(const u64* __restrict__ Vraw, u64* __restrict__ V, u64 L, u64 lw, u64 nwords)
{
    for (u64 w = blockIdx.x * (u64)blockDim.x + threadIdx.x; w < nwords; w += (u64)gridDim.x * blockDim.x) {
        u64 bits = 0;
        for (int i = 0; i < 64; i++) {
            u64 c = w * 64 + i;
            if (c >= L) break;
            u64 q = c + c / lw;
            bits |= ((Vraw[q >> 6] >> (q & 63)) & 1ull) << i;
        }
        V[w] = bits;
    }
}

// ---------------------------------------------------------------------------------------------
// K2: runs and segments.  build_eds_boundaries :101-115, build_leds_boundaries :133-190.
// ---------------------------------------------------------------------------------------------
__global__ void k_runstart_words(const u64* __restrict__ V, u64* __restrict__ H, u64* __restrict__ cnt,
                                 u64 L, u64 nwords)
{
    for (u64 w = blockIdx.x * (u64)blockDim.x + threadIdx.x; w < nwords; w += (u64)gridDim.x * blockDim.x) {
        u64 v = V[w];
        u64 carry = w ? (V[w - 1] >> 63) : ((~v) & 1ull);     // forces a run start at column 0
        u64 hbits = v ^ ((v << 1) | carry);
        u64 rem = L - w * 64;
        if (rem < 64) hbits &= (1ull << rem) - 1ull;
        H[w] = hbits;
        cnt[w] = __builtin_popcountll(hbits);
    }
}

__global__ void k_write_positions(const u64* __restrict__ H, const u64* __restrict__ wbase,
                                  u64* __restrict__ pos, u64 nwords, const u64* __restrict__ total, u64 L)
{
    for (u64 w = blockIdx.x * (u64)blockDim.x + threadIdx.x; w < nwords; w += (u64)gridDim.x * blockDim.x) {
        u64 hbits = H[w];
        u64 o = wbase[w];
        while (hbits) {
            pos[o++] = w * 64 + __builtin_ctzll(hbits);
            hbits &= hbits - 1;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) pos[*total] = L;
}

// flag[r] = 1 iff run r starts an l-EDS segment
__global__ void k_seg_flags(const u64* __restrict__ run_start, const u64* __restrict__ V,
                            const u64* __restrict__ R_ptr, u64 l, u64* __restrict__ flag)
{
    const u64 R = *R_ptr;
    for (u64 r = blockIdx.x * (u64)blockDim.x + threadIdx.x; r < R; r += (u64)gridDim.x * blockDim.x) {
        u64 a = run_start[r], b = run_start[r + 1];
        u32 var = (u32)(V[a >> 6] >> (a & 63)) & 1u;
        u64 fl;
        if (!var) fl = (b - a >= l) || r == 0 || r == R - 1;             // standalone common run
        else if (r == 0) fl = 1;
        else {
            u64 pa = run_start[r - 1];
            fl = (a - pa >= l) || (r - 1 == 0);                          // previous common standalone
        }
        flag[r] = fl;
    }
}

__global__ void k_write_segs(const u64* __restrict__ run_start, const u64* __restrict__ flag,
                             const u64* __restrict__ sidx, const u64* __restrict__ R_ptr,
                             const u64* __restrict__ nseg_ptr, u64* __restrict__ seg_start,
                             u64* __restrict__ Hseg, u64 L)
{
    const u64 R = *R_ptr;
    for (u64 r = blockIdx.x * (u64)blockDim.x + threadIdx.x; r < R; r += (u64)gridDim.x * blockDim.x) {
        if (flag[r]) {
            u64 a = run_start[r];
            seg_start[sidx[r]] = a;
            atomicOr(&Hseg[a >> 6], 1ull << (a & 63));
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) seg_start[*nseg_ptr] = L;
}

__global__ void k_popc_words(const u64* __restrict__ H, u64* __restrict__ cnt, u64 nwords)
{
    for (u64 w = blockIdx.x * (u64)blockDim.x + threadIdx.x; w < nwords; w += (u64)gridDim.x * blockDim.x)
        cnt[w] = __builtin_popcountll(H[w]);
}

// ---------------------------------------------------------------------------------------------
// K3/K5b shared: group the rows of one variant segment by their gap-stripped string.
//   msa_transforms.cpp:262-293 — strings in order of first appearance, ids ascending.
// Workgroup-level (GT threads).  LDS carve (18 bytes per row):
//   key[S] u64 | rep_row[S] u32 | run[S] u32 | gid[S] u16
// ---------------------------------------------------------------------------------------------
constexpr int GT = 512;
// BIG (more rows than the LDS holds, MsaPipeline::LDS_ROWS): the same arrays in a per-workgroup slice of HBM scratch,
// group ids of 32 bits (a segment may have more than 65535 distinct strings).
template <bool BIG> struct SegLdsT {
    using gid_t = std::conditional_t<BIG, u32, uint16_t>;
    static constexpr u32 NONE = BIG ? 0xffffffffu : 0xffffu;
    u64* key; u32* rep_row; u32* run; gid_t* gid;
    __device__ SegLdsT(uint8_t* base, u32 S)
    {
        key = reinterpret_cast<u64*>(base);
        rep_row = reinterpret_cast<u32*>(base + (size_t)8 * S);
        run = reinterpret_cast<u32*>(base + (size_t)12 * S);
        gid = reinterpret_cast<gid_t*>(base + (size_t)16 * S);
    }
    __host__ __device__ static size_t bytes(u32 S) { return (size_t)(16 + sizeof(gid_t)) * S; }
};
// tables of the generic emitter's .seds walk (k_emit_variant): per 64-row block the bytes / starts (u32) and ids of its
// groups, per row its offset inside its group's part of the block (u16) and the index of its group in the block (u8),
// per block the number of groups (u8)
template <bool BIG> __host__ __device__ inline size_t walk_table_bytes(u32 S)
{
    const size_t nblk = (S + 63u) >> 6, S2 = (S + 1u) & ~1u;
    return nblk * (256 + 64 * sizeof(typename SegLdsT<BIG>::gid_t)) + S2 * 3 + nblk + 16;
}

// Cells of one segment.  Read from HBM a cell costs a chain of dependent loads (V word, slot table,
// vc byte); segments of up to STAGE_COLS pure variant columns are first copied into LDS (`st`), which
// turns the generic kernels' latency-bound row walks into LDS reads.
constexpr u32 STAGE_COLS = 64;            // variant columns of a segment that are staged at most
constexpr u32 STAGE_WMAX = 512;           // widest segment (variant + common columns) that is staged
// staging area: slot_tab[cap] u64 | cmap[STAGE_WMAX] u8 (staged column of segment column c, 0xFF: a common column)
// | cref[STAGE_WMAX] u8 (its reference byte) | the staged variant columns (pitch Spad)
__host__ __device__ inline u32 stage_cols_offset(u32 cap) { return (cap * 8u + 2u * STAGE_WMAX + 15u) & ~15u; }
struct SegCells {
    const MsaView& mv; u64 a; const uint8_t* st; u32 cap;
    __device__ __forceinline__ u32 at(u64 c, u32 r) const
    {
        // (the staging area is LDS: an explicit LDS pointer, or the reads become flat loads - `st` is a select of an LDS
        // address and nullptr, whose address space the compiler does not follow)
        typedef const __attribute__((address_space(3))) uint8_t* lds_bytes;
        if (st && (cap >> 31)) return ((lds_bytes)st)[stage_cols_offset(cap & 0xffffu) + (u32)(c - a) * mv.Spad + r];   // every column staged in place
        if (st) {                                             // variant columns + column map
            const lds_bytes sl = (lds_bytes)st;
            const u32 i = sl[cap * 8u + (u32)(c - a)];
            return i == 0xffu ? sl[cap * 8u + STAGE_WMAX + (u32)(c - a)] : sl[stage_cols_offset(cap) + i * mv.Spad + r];
        }
        return mv.vbit(c) ? mv.vc[mv.slot(c) * mv.Spad + r] : mv.ref_byte(c);
    }
};
// all threads of the workgroup; returns the staging area with the segment's cells or nullptr (too wide, or more variant
// columns than fit).  A segment of at most cap_cols columns is staged column for column (a common column inside - context
// merge - as a splat of its reference byte; `cap` comes back with bit 31 set: direct indexing); a wider one keeps only its
// variant columns plus a column map.
__device__ const uint8_t* stage_columns(const MsaView& mv, u64 a, u64 b, uint8_t* buf, u32& cap, u32* flag_sh)
{
    const u64 ncol = b - a;
    const u32 cap_cols = cap;
    if (!buf || ncol > STAGE_WMAX) return nullptr;
    u64* slot_tab = reinterpret_cast<u64*>(buf);
    if (ncol <= cap_cols) {
        uint8_t* cols = buf + stage_cols_offset(cap_cols);
        __syncthreads();                                      // (the previous segment is done with the staging area)
        if (threadIdx.x < ncol) {
            const u64 c = a + threadIdx.x;
            slot_tab[threadIdx.x] = mv.vbit(c) ? mv.slot(c) : ((1ull << 63) | mv.ref_byte(c));
        }
        __syncthreads();
        const u32 vec = mv.Spad / 16;                         // Spad % 16 == 0
        for (u32 i = threadIdx.x; i < (u32)ncol * vec; i += blockDim.x) {
            const u32 c = i / vec, o = (i - c * vec) * 16;
            const u64 sl = slot_tab[c];
            uint4 v;
            if (sl >> 63) { const u32 b4 = (u32)(sl & 0xffu) * 0x01010101u; v = make_uint4(b4, b4, b4, b4); }
            else v = *reinterpret_cast<const uint4*>(mv.vc + sl * (u64)mv.Spad + o);
            *reinterpret_cast<uint4*>(cols + (size_t)c * mv.Spad + o) = v;
        }
        __syncthreads();
        cap = cap_cols | 0x80000000u;
        return buf;
    }
    uint8_t* cmap = buf + cap_cols * 8u;
    uint8_t* cref = cmap + STAGE_WMAX;
    uint8_t* cols = buf + stage_cols_offset(cap_cols);
    __syncthreads();                                          // (the previous segment is done with the staging area)
    if (threadIdx.x == 0) *flag_sh = 0;
    __syncthreads();
    for (u32 c = threadIdx.x; c < (u32)ncol; c += blockDim.x) {
        if (mv.vbit(a + c)) {
            const u32 idx = atomicAdd(flag_sh, 1u);
            if (idx < cap_cols) { slot_tab[idx] = mv.slot(a + c); cmap[c] = (uint8_t)idx; }
        } else { cmap[c] = 0xff; cref[c] = (uint8_t)mv.ref_byte(a + c); }
    }
    __syncthreads();
    const u32 nvar = *flag_sh;
    if (nvar > cap_cols) return nullptr;                      // (workgroup-uniform)
    const u32 vec = mv.Spad / 16;                             // Spad % 16 == 0
    for (u32 i = threadIdx.x; i < nvar * vec; i += blockDim.x) {
        const u32 c = i / vec, o = (i - c * vec) * 16;
        *reinterpret_cast<uint4*>(cols + (size_t)c * mv.Spad + o) =
            *reinterpret_cast<const uint4*>(mv.vc + slot_tab[c] * (u64)mv.Spad + o);
    }
    __syncthreads();
    return buf;
}

// Do rows r1 and r2 spell the same gap-stripped string over [a,b)?  Written for a wave whose lanes
// compare different row pairs: the common case (the two rows are byte-identical) is one pass with no
// data-dependent branch; otherwise one merged two-pointer loop in which every lane advances at least
// one of its pointers per iteration (nested skip loops diverge lane by lane: measured 200 K cycles
// per call on a 32-column segment).
__device__ bool seg_rows_equal(const SegCells& sc, u64 a, u64 b, u32 r1, u32 r2)
{
    u32 diff = 0, seen0 = 0;
    for (u64 c = a; c < b; c++) {
        const u32 x = sc.at(c, r1), y = sc.at(c, r2);
        diff |= x ^ y;
        seen0 |= (x == 0) | (y == 0);
    }
    if (!diff && !seen0) return true;
    u64 c1 = a, c2 = a;
    while (true) {
        u32 x = c1 < b ? sc.at(c1, r1) : 0u, y = c2 < b ? sc.at(c2, r2) : 0u;
        if (x == 0) c1 = b;                  // '\0' ends the row (msa_transforms.cpp:282)
        if (y == 0) c2 = b;
        const bool s1 = x == '-' || x == '\n', s2 = y == '-' || y == '\n';
        if (s1) c1++;
        if (s2) c2++;
        if (!s1 && !s2) {
            if (x != y) return false;
            if (x == 0) return true;         // both exhausted
            c1++; c2++;
        }
    }
}

__device__ u32 seg_row_len(const SegCells& sc, u64 a, u64 b, u32 r)
{
    u32 len = 0;
    for (u64 c = a; c < b; c++) {
        u32 ch = sc.at(c, r);
        if (ch == 0) break;
        if (ch != '-' && ch != '\n') len++;
    }
    return len;
}

// hash-table grouping (S <= HT_MAX_ROWS): every row inserts its key into an LDS open-addressing
// table and takes atomicMin(row) on its slot, so each group learns its first row in O(1) rounds
// (the iterative path below needs one barrier round per distinct string).  Hashed keys are verified
// byte for byte against the group's first row; a collision falls back to the iterative path.
constexpr u32 HT_MAX_ROWS = 2048, HT_SIZE = 4096;
// table entries for S rows: twice the rows, a power of two (1024 rows -> 2048 entries: two workgroups fit a CU)
__host__ __device__ inline u32 ht_size_of(u32 S) { u32 n = 256; while (n < 2u * S) n <<= 1; return n < HT_SIZE ? n : HT_SIZE; }
// one u32 per entry: while the rows insert themselves it holds the row that claimed the entry (keys are compared through
// lds.key[]), afterwards the first row of the entry's string
struct HtLds {
    u32* tabm; u32* bm; u32* pre; u32* flag;
    __device__ HtLds(uint8_t* base, u32 hsz)
    {
        tabm = reinterpret_cast<u32*>(base);
        bm = reinterpret_cast<u32*>(base + (size_t)4 * hsz);
        pre = bm + 72;
        flag = pre + 72;
    }
    __host__ __device__ static size_t bytes(u32 hsz) { return (size_t)4 * hsz + 4 * (72 + 72 + 8); }
};

// returns k (number of distinct strings); fills lds.gid[], lds.rep_row[0..k)
template <bool BIG>
__device__ u32 group_segment(const MsaView& mv, u64 a, u64 b, SegLdsT<BIG>& lds, u32* rep_sh, const uint8_t* st, u32 cap)
{
    using gid_t = typename SegLdsT<BIG>::gid_t;
    constexpr u32 GID_NONE = SegLdsT<BIG>::NONE;
    const u32 S = mv.S;
    const SegCells sc{mv, a, st, cap};
    const bool exact = (b - a) <= 8;
    u32 saw_nl = 0;
    // keys of R rows per thread side by side, column by column (R = 1, 2 or 4 by the row count): no branch depends on a
    // cell (what a column is - staged, mapped, a common column's reference byte - is decided once per column for the R
    // rows; '-', '\n' and the end of a row at '\0' are selects), where a loop per row with its `break` ran ~80 mostly
    // scalar instructions per cell
    auto make_keys = [&](auto rc) {
        constexpr int R = decltype(rc)::value;
        for (u32 r0 = threadIdx.x; r0 < S; r0 += R * GT) {
            u32 rr[R], len[R], ended[R];
            u64 key[R];
#pragma unroll
            for (int i = 0; i < R; i++) {
                const u32 r = r0 + (u32)i * GT;
                rr[i] = r < S ? r : S - 1; len[i] = 0; ended[i] = 0; key[i] = exact ? 0ull : 0xcbf29ce484222325ull;
            }
            for (u64 c = a; c < b; c++) {
#pragma unroll
                for (int i = 0; i < R; i++) {
                    const u32 ch = sc.at(c, rr[i]);
                    ended[i] |= ch == 0 ? 1u : 0u;             // '\0' ends the row (msa_transforms.cpp:282)
                    const bool nl = ch == '\n', keep = !ended[i] && ch != '-' && !nl;
                    saw_nl |= (nl && !ended[i]) ? 1u : 0u;
                    if (exact) key[i] |= keep ? (u64)ch << (8 * len[i]) : 0ull;
                    else key[i] = keep ? (key[i] ^ ch) * 0x100000001b3ull : key[i];
                    len[i] += keep ? 1u : 0u;
                }
            }
#pragma unroll
            for (int i = 0; i < R; i++) {
                const u32 r = r0 + (u32)i * GT;
                if (r < S) {
                    lds.key[r] = exact ? key[i] : (key[i] ^ len[i]) * 0x100000001b3ull;
                    lds.gid[r] = GID_NONE;
                }
            }
        }
    };
    if (S <= (u32)GT) make_keys(std::integral_constant<int, 1>{});
    else if (S <= 2u * GT) make_keys(std::integral_constant<int, 2>{});
    else make_keys(std::integral_constant<int, 4>{});
    if (saw_nl) atomicOr(&mv.hdr->status, (u64)(ST_LAYOUT | ST_NEWLINE_IN_DATA));   // a row is ragged

    if (!BIG && S <= HT_MAX_ROWS) {
        const u32 hsz = ht_size_of(S);
        HtLds ht(reinterpret_cast<uint8_t*>(lds.gid + ((S + 7) & ~7u)), hsz);
        for (u32 i = threadIdx.x; i < hsz; i += GT) ht.tabm[i] = 0xffffffffu;
        for (u32 i = threadIdx.x; i < 72; i += GT) ht.bm[i] = 0;
        if (threadIdx.x == 0) *ht.flag = 0;
        __syncthreads();
        for (u32 r = threadIdx.x; r < S; r += GT) {
            const u64 kk = lds.key[r];
            u32 slot = (u32)(mix64(kk) >> 20) & (hsz - 1);
            while (true) {
                const u32 cur = atomicCAS(&ht.tabm[slot], 0xffffffffu, r);
                if (cur == 0xffffffffu || lds.key[cur] == kk) { lds.run[r] = slot; break; }
                slot = (slot + 1) & (hsz - 1);
            }
        }
        __syncthreads();
        for (u32 i = threadIdx.x; i < hsz; i += GT) ht.tabm[i] = 0xffffffffu;
        __syncthreads();
        for (u32 r = threadIdx.x; r < S; r += GT) atomicMin(&ht.tabm[lds.run[r]], r);
        __syncthreads();
        for (u32 r = threadIdx.x; r < S; r += GT) {
            const u32 f = ht.tabm[lds.run[r]];
            if (!exact && f != r && !seg_rows_equal(sc, a, b, r, f)) *ht.flag = 1;
            lds.rep_row[r] = f;                       // temporarily: first row of r's group
            if (f == r) atomicOr(&ht.bm[r >> 5], 1u << (r & 31));
        }
        __syncthreads();
        if (*ht.flag == 0) {
            const u32 nwords = (S + 31) >> 5;                 // <= 64: one wave scans the first-row counts of the words
            if (threadIdx.x < 64) {
                const u32 c = threadIdx.x < nwords ? (u32)__builtin_popcount(ht.bm[threadIdx.x]) : 0u;
                const u32 incl = wave_scan_incl(c);
                if (threadIdx.x < nwords) ht.pre[threadIdx.x] = incl - c;
                if (threadIdx.x == 63) *rep_sh = incl;
            }
            __syncthreads();
            u32 myg[HT_MAX_ROWS / GT];
            for (u32 j = 0, r = threadIdx.x; r < S; r += GT, j++) {
                const u32 f = lds.rep_row[r];
                myg[j] = ht.pre[f >> 5] + __builtin_popcount(ht.bm[f >> 5] & ((1u << (f & 31)) - 1u));
            }
            __syncthreads();                              // rep_row[] is rewritten below
            for (u32 j = 0, r = threadIdx.x; r < S; r += GT, j++) {
                const u32 f = lds.rep_row[r];
                lds.gid[r] = (gid_t)myg[j];
                (void)f;
            }
            __syncthreads();
            for (u32 j = 0, r = threadIdx.x; r < S; r += GT, j++)
                if (ht.tabm[lds.run[r]] == r) lds.rep_row[myg[j]] = r;
            const u32 k = *rep_sh;
            __syncthreads();
            return k;
        }
        __syncthreads();                                  // hash collision: redo iteratively
    }

    u32 g = 0;
    u32 cursor = threadIdx.x;                       // first possibly unassigned row of this thread
    while (true) {
        if (threadIdx.x == 0) *rep_sh = 0xffffffffu;
        __syncthreads();
        while (cursor < S && lds.gid[cursor] != GID_NONE) cursor += GT;
        if (cursor < S) atomicMin(rep_sh, cursor);
        __syncthreads();
        const u32 rep = *rep_sh;
        if (rep == 0xffffffffu) break;
        const u64 rk = lds.key[rep];
        for (u32 r = cursor; r < S; r += GT) {
            if (lds.gid[r] == GID_NONE && lds.key[r] == rk &&
                (exact || r == rep || seg_rows_equal(sc, a, b, r, rep)))
                lds.gid[r] = (gid_t)g;
        }
        if (threadIdx.x == 0) lds.rep_row[g] = rep;
        g++;
        __syncthreads();
    }
    return g;
}

struct SegParams {
    MsaView mv; const u64* seg_start; const u64* nseg_ptr; u64* eds_len; u64* seds_len; u64 tok_total;
    const u64* list; const u64* list_n;       // when set: only these segments (left over by the fast path)
    u32 stage_cols = 0, stage_off = 0;        // LDS column staging: capacity in columns, byte offset in the dynamic LDS
    // grouping cache (count -> emit): item `it` of the list (or segment `it`) keeps k, its group ids and first rows
    uint8_t* gcache = nullptr; u64 gcache_cap = 0; u64 gcache_stride = 0;
    u64* long_list = nullptr; u64* long_count = nullptr;   // common segments for k_emit_common_long (see common_is_long)
    uint8_t* scratch = nullptr; u64 scratch_stride = 0;    // BIG: per-workgroup slice of HBM for the row tables
};
// cache entry: u32 k, pad; gid[S (rounded up to 8)] (u16, BIG: u32); u32 rep_row[S]
__host__ __device__ inline u64 gcache_stride_of(u32 S, bool big) { return 16ull + (u64)((S + 7u) & ~7u) * (big ? 4u : 2u) + (u64)S * 4u; }

// ---- common segments (msa_transforms.cpp:245-258: "{" + the reference row's text + "}" and "{0}") ---------------------
// A thread per common segment (they alternate with the variant ones): four coalesced table reads, then the few
// reference bytes as unaligned 16-byte copies.  Segments longer than LONG_COMMON columns are left to k_emit_common_long
// (a workgroup per segment, all workgroups for the very long ones).
constexpr u64 LONG_COMMON = 512, HUGE_COMMON = 1u << 20;
template <int N> struct __attribute__((packed, aligned(1))) PackedBytes { uint8_t b[N]; };
template <int N> __device__ __forceinline__ void store_small(uint8_t* p, u64 v)   // the low N (2, 4, 8) bytes of v, any alignment
{
    PackedBytes<N> t;
    __builtin_memcpy(&t, &v, N);
    __builtin_memcpy(p, &t, N);
}
__device__ __forceinline__ bool common_is_long(u64 ncol) { return ncol > LONG_COMMON; }

__global__ void __launch_bounds__(256) k_emit_common_seg(MsaView mv, const u64* __restrict__ seg_start, const u64* __restrict__ nseg_ptr,
                                                         const u64* __restrict__ eds_off, const u64* __restrict__ seds_off,
                                                         uint8_t* __restrict__ eds, uint8_t* __restrict__ seds)
{
    const u64 nseg = *nseg_ptr, p0 = mv.vbit(0) ? 0 : 1;
    const uint8_t* row0 = mv.file + mv.row_start[0];
    for (u64 seg = (1 - p0) + 2 * (blockIdx.x * (u64)blockDim.x + threadIdx.x); seg < nseg; seg += 2 * (u64)gridDim.x * blockDim.x) {
        const u64 a = seg_start[seg], clen = seg_start[seg + 1] - a;
        if (common_is_long(clen)) continue;
        uint8_t* e = eds + eds_off[seg];
        uint8_t* q = seds + seds_off[seg];
        store_small<2>(q, (u32)'{' | ((u32)'0' << 8)); q[2] = '}';
        e[clen + 1] = '}';
        if (mv.lw == 0 && clen >= 16) {
            e[0] = '{';
            for (u64 o = 0; o < clen; o += 16) {
                const u64 oo = o + 16 <= clen ? o : clen - 16;        // the last piece ends with the segment (it overlaps the one before)
                store16u(e + 1 + oo, load16u(row0 + a + oo));
            }
        } else if (mv.lw == 0) {
            // "{" + up to 15 letters: one 16-byte load (row 0 is followed by more of the file: never past its end), then
            // the clen + 1 bytes as 16 / 8 + 4 + 2 + 1 byte stores
            const uint4 v = load16u(row0 + a);
            u64 lo = ((u64)v.y << 32) | v.x, hi = ((u64)v.w << 32) | v.z;
            hi = (hi << 8) | (lo >> 56); lo = (lo << 8) | (u64)'{';
            const u32 m = (u32)clen + 1u;
            if (m == 16u) store16u(e, make_uint4((u32)lo, (u32)(lo >> 32), (u32)hi, (u32)(hi >> 32)));
            else {
                uint8_t* d = e;
                if (m & 8u) { store_small<8>(d, lo); d += 8; lo = hi; }
                if (m & 4u) { store_small<4>(d, lo); d += 4; lo >>= 32; }
                if (m & 2u) { store_small<2>(d, lo); d += 2; lo >>= 16; }
                if (m & 1u) *d = (uint8_t)lo;
            }
        } else {
            e[0] = '{';
            for (u64 o = 0; o < clen; o++) e[1 + o] = (uint8_t)mv.ref_byte(a + o);
        }
    }
}

// the long ones: a workgroup per listed segment; every workgroup takes its share of a segment of more than HUGE_COMMON columns
__global__ void __launch_bounds__(256) k_emit_common_long(MsaView mv, const u64* __restrict__ seg_start, const u64* __restrict__ eds_off,
                                                          const u64* __restrict__ seds_off, const u64* __restrict__ list,
                                                          const u64* __restrict__ list_n, uint8_t* __restrict__ eds, uint8_t* __restrict__ seds)
{
    const u64 n = *list_n;
    const uint8_t* row0 = mv.file + mv.row_start[0];
    for (u64 i = 0; i < n; i++) {
        const u64 seg = list[i], a = seg_start[seg], clen = seg_start[seg + 1] - a;
        const bool huge = clen > HUGE_COMMON;
        if (!huge && i % gridDim.x != blockIdx.x) continue;
        const u64 t = huge ? blockIdx.x * (u64)blockDim.x + threadIdx.x : threadIdx.x, nt = huge ? (u64)gridDim.x * blockDim.x : blockDim.x;
        uint8_t* e = eds + eds_off[seg];
        if (t == 0) {
            e[0] = '{'; e[clen + 1] = '}';
            uint8_t* q = seds + seds_off[seg];
            q[0] = '{'; q[1] = '0'; q[2] = '}';
        }
        if (mv.lw == 0) {
            for (u64 o = 16 * t; o < clen; o += 16 * nt) {               // (clen > LONG_COMMON >= 16)
                const u64 oo = o + 16 <= clen ? o : clen - 16;
                store16u(e + 1 + oo, load16u(row0 + a + oo));
            }
        } else
            for (u64 o = t; o < clen; o += nt) e[1 + o] = (uint8_t)mv.ref_byte(a + o);
    }
}

// K3: per-segment output sizes.  common: "{" ref "}" and "{0}"; variant: see generate_output.
template <bool BIG>
__global__ void __launch_bounds__(GT) k_seg_count(SegParams p)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    __shared__ u32 rep_sh;
    __shared__ u64 sum_sh;
    using gid_t = typename SegLdsT<BIG>::gid_t;
    SegLdsT<BIG> lds(BIG ? p.scratch + blockIdx.x * p.scratch_stride : lds_raw, p.mv.S);
    if (p.mv.hdr->status) return;                     // vc overflow: the host grows vc and replans
    // without a work list: every segment.  Variant and common segments alternate (item `it` = the it-th variant segment);
    // the common ones are a thread each
    const u64 nseg = *p.nseg_ptr, p0 = p.mv.vbit(0) ? 0 : 1;
    if (!p.list)
        for (u64 seg = (1 - p0) + 2 * (blockIdx.x * (u64)GT + threadIdx.x); seg < nseg; seg += 2 * (u64)gridDim.x * GT) {
            const u64 ncol = p.seg_start[seg + 1] - p.seg_start[seg];
            p.eds_len[seg] = 2 + ncol;
            p.seds_len[seg] = 3;
            if (common_is_long(ncol)) p.long_list[atomicAdd(p.long_count, 1ull)] = seg;
        }
    const u64 nitems = p.list ? *p.list_n : (nseg > p0 ? (nseg - p0 + 1) / 2 : 0);
    for (u64 it = blockIdx.x; it < nitems; it += gridDim.x) {
        const u64 seg = p.list ? p.list[it] : p0 + 2 * it;
        const u64 a = p.seg_start[seg], b = p.seg_start[seg + 1];
        if (threadIdx.x == 0) sum_sh = 0;
        u32 cap = p.stage_cols;
        const uint8_t* st = stage_columns(p.mv, a, b, p.stage_cols ? lds_raw + p.stage_off : nullptr, cap, &rep_sh);
        const u32 k = group_segment(p.mv, a, b, lds, &rep_sh, st, cap);
        if (it < p.gcache_cap) {                             // the emitter takes the grouping from here
            uint8_t* ce = p.gcache + it * (u64)p.gcache_stride;
            gid_t* cg = reinterpret_cast<gid_t*>(ce + 16);
            u32* cr = reinterpret_cast<u32*>(ce + 16 + (size_t)((p.mv.S + 7u) & ~7u) * sizeof(gid_t));
            if (threadIdx.x == 0) *reinterpret_cast<u32*>(ce) = k;
            for (u32 r = threadIdx.x; r < p.mv.S; r += GT) cg[r] = lds.gid[r];
            for (u32 g = threadIdx.x; g < k; g += GT) cr[g] = lds.rep_row[g];
        }
        const SegCells sc{p.mv, a, st, cap};
        u64 mine = 0;
        for (u32 g = threadIdx.x; g < k; g += GT) mine += seg_row_len(sc, a, b, lds.rep_row[g]);
        if (mine) atomicAdd(&sum_sh, mine);
        __syncthreads();
        if (threadIdx.x == 0) {
            p.eds_len[seg] = 2 + (u64)(k - 1) + sum_sh;
            p.seds_len[seg] = (u64)k + p.tok_total;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// K5: parameters of the generic emitter
// ---------------------------------------------------------------------------------------------
struct EmitParams {
    MsaView mv; const u64* seg_start; const u64* nseg_ptr; const u64* Hseg; const u64* segbase;
    const u64* eds_off; const u64* seds_off; uint8_t* eds; uint8_t* seds; u64 nwords;
    const u64* list; const u64* list_n;
    u32 stage_cols = 0, stage_off = 0;
    const uint8_t* gcache = nullptr; u64 gcache_cap = 0; u64 gcache_stride = 0;      // see SegParams
    uint8_t* scratch = nullptr; u64 scratch_stride = 0;
    const u64* list2 = nullptr; const u64* list2_n = nullptr; const uint8_t* gcache2 = nullptr;   // a second work list behind the first
};

// ---------------------------------------------------------------------------------------------
// K5b: variant-segment text.  msa_transforms.cpp:297-317.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void write_decimal(uint8_t* dst, u32 v, u32 nd)
{
    for (int i = (int)nd - 1; i >= 0; i--) { dst[i] = (uint8_t)('0' + v % 10u); v /= 10u; }
}

template <bool BIG>
__global__ void __launch_bounds__(GT) k_emit_variant(EmitParams p)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    __shared__ u32 rep_sh;
    const MsaView& mv = p.mv;
    const u32 S = mv.S;
    using gid_t = typename SegLdsT<BIG>::gid_t;
    uint8_t* const tables = BIG ? p.scratch + blockIdx.x * p.scratch_stride : lds_raw;
    SegLdsT<BIG> lds(tables, S);
    if (mv.hdr->status) return;
    const u64 nseg = *p.nseg_ptr, p0 = mv.vbit(0) ? 0 : 1;      // items as in k_seg_count
    const u64 n1 = p.list ? *p.list_n : (nseg > p0 ? (nseg - p0 + 1) / 2 : 0);
    const u64 nitems = n1 + (p.list2 ? *p.list2_n : 0);
    const u32 lane = threadIdx.x & 63;
    for (u64 it0 = blockIdx.x; it0 < nitems; it0 += gridDim.x) {
        const bool second = it0 >= n1;                       // (workgroup-uniform)
        const u64 it = second ? it0 - n1 : it0;
        const u64 seg = second ? p.list2[it] : p.list ? p.list[it] : p0 + 2 * it;
        const u64 a = p.seg_start[seg], b = p.seg_start[seg + 1];
        u32 cap = p.stage_cols;
        const uint8_t* st = stage_columns(mv, a, b, p.stage_cols ? lds_raw + p.stage_off : nullptr, cap, &rep_sh);
        u32 k;
        if (it < p.gcache_cap) {                             // grouped by k_seg_count already
            const uint8_t* ce = (second ? p.gcache2 : p.gcache) + it * (u64)p.gcache_stride;
            const gid_t* cg = reinterpret_cast<const gid_t*>(ce + 16);
            const u32* cr = reinterpret_cast<const u32*>(ce + 16 + (size_t)((S + 7u) & ~7u) * sizeof(gid_t));
            k = *reinterpret_cast<const u32*>(ce);
            __syncthreads();                                  // (the previous segment's readers of gid / rep_row are done)
            for (u32 r = threadIdx.x; r < S; r += GT) lds.gid[r] = cg[r];
            for (u32 g = threadIdx.x; g < k; g += GT) lds.rep_row[g] = cr[g];
            __syncthreads();
        } else k = group_segment(mv, a, b, lds, &rep_sh, st, cap);
        const SegCells sc{mv, a, st, cap};
        uint8_t* eds = p.eds + p.eds_off[seg];
        uint8_t* seds = p.seds + p.seds_off[seg];

        // ---- eds: "{" s0 "," s1 ... "}" ; key[] is reused for the string offsets
        u64* goff = lds.key;
        for (u32 g = threadIdx.x; g < k; g += GT) {
            goff[g] = seg_row_len(sc, a, b, lds.rep_row[g]);
            lds.run[g] = 0;
        }
        __syncthreads();
        // token bytes per group: sum over member rows of digits(r+1)+1
        for (u32 r = threadIdx.x; r < S; r += GT) atomicAdd(&lds.run[lds.gid[r]], ndigits(r + 1) + 1);
        __syncthreads();
        if (threadIdx.x == 0) {
            u64 eo = 1;
            u32 so = 0;
            for (u32 g = 0; g < k; g++) {
                u64 len = goff[g]; goff[g] = eo; eo += len + 1;
                u32 t = lds.run[g]; lds.run[g] = so + 1; so += 1 + t;
            }
            eds[0] = '{';
        }
        __syncthreads();
        for (u32 g = threadIdx.x; g < k; g += GT) {
            uint8_t* dst = eds + goff[g];
            const u32 r = lds.rep_row[g];
            for (u64 c = a; c < b; c++) {
                u32 ch = sc.at(c, r);
                if (ch == 0) break;
                if (ch != '-' && ch != '\n') *dst++ = (uint8_t)ch;
            }
            *dst = (g + 1 < k) ? ',' : '}';
            seds[lds.run[g] - 1] = '{';
        }
        __syncthreads();

        // ---- seds: run[g] = next write offset of group g.  A row's token goes behind the tokens of the earlier rows of its
        // group.  When the staging area (free again: the .eds text is written) holds the tables, all waves of the workgroup work on
        // it: (A) every wave takes its share of the 64-row blocks and finds, per distinct group of the block, the bytes of its
        // rows and every row's offset among them; (B) wave 0 walks the blocks in order and turns the per-block group
        // totals into start offsets (a block's groups are distinct: one lane each); (C) all waves store their tokens.
        // Otherwise (no staging area: very many rows) wave 0 walks the rows block by block.
        const u32 nblk = (S + 63u) >> 6, S2 = (S + 1u) & ~1u, wv = uniform32(threadIdx.x >> 6);
        const size_t walk_bytes = walk_table_bytes<BIG>(S);
        // (BIG: the tables live behind the row tables in the workgroup's slice of HBM scratch)
        const bool par = BIG || (p.stage_cols != 0 && walk_bytes <= (size_t)p.stage_cols * (8 + (size_t)mv.Spad));
        auto token_of = [&](u32 r, u32 tl) -> u64 {            // "ddd," little-endian: first digit in byte 0
            u64 tok = (u64)',' << (8 * (tl - 1));
            u32 v = r + 1;
            for (int i = (int)tl - 2; i >= 0; i--) { tok |= (u64)('0' + v % 10u) << (8 * i); v /= 10u; }
            return tok;
        };
        auto store_token = [&](uint8_t* dst, u64 tok, u32 tl) {
            dst[0] = (uint8_t)tok; dst[1] = (uint8_t)(tok >> 8);
            if (tl >= 3) dst[2] = (uint8_t)(tok >> 16);
            if (tl >= 4) dst[3] = (uint8_t)(tok >> 24);
            if (tl >= 5) dst[4] = (uint8_t)(tok >> 32);
            if (tl >= 6) for (u32 i = 5; i < tl; i++) dst[i] = (uint8_t)(tok >> (8 * i));
        };
        if (par) {
            uint8_t* wb = BIG ? tables + ((SegLdsT<BIG>::bytes(S) + 15) & ~(size_t)15) : lds_raw + p.stage_off;
            constexpr size_t PER_BLK = 256 + 64 * sizeof(gid_t);
            u32* LT = reinterpret_cast<u32*>(wb);                                  // [block][i]: bytes of the block's i-th group, then its start
            gid_t* LG = reinterpret_cast<gid_t*>(wb + (size_t)nblk * 256);         // [block][i]: that group
            uint16_t* REL = reinterpret_cast<uint16_t*>(wb + (size_t)nblk * PER_BLK);  // [row]: bytes of the earlier rows of its group in its block
            uint8_t* IDX = wb + (size_t)nblk * PER_BLK + (size_t)S2 * 2;           // [row]: index of its group in its block's list
            uint8_t* NG = IDX + S2;                                                // [block]: distinct groups
            for (u32 blk = wv; blk < nblk; blk += GT / 64) {                       // (A)
                const u32 r = blk * 64u + lane;
                const bool valid = r < S;
                const u32 g = valid ? lds.gid[r] : 0xffffffffu;
                const u32 tl = ndigits(r + 1) + 1;
                const u32 tlA = __shfl(tl, 0, 64);              // <= 2 token lengths per 64 rows
                const u64 maskA = ballot64(valid && tl == tlA);
                u32 myrel = 0, myidx = 0, idx = 0;
                u64 todo = ballot64(valid);
                while (todo) {
                    const int leader = __builtin_ctzll(todo);
                    const u32 g0 = (u32)__builtin_amdgcn_readlane((int)g, leader);
                    const u64 m = ballot64(valid && g == g0);
                    if (valid && g == g0) { myrel = mbcnt(m & maskA) * tlA + mbcnt(m & ~maskA) * (tlA + 1); myidx = idx; }
                    if (lane == (u32)leader) {
                        LG[blk * 64u + idx] = (gid_t)g0;
                        LT[blk * 64u + idx] = (u32)__builtin_popcountll(m & maskA) * tlA + (u32)__builtin_popcountll(m & ~maskA) * (tlA + 1);
                    }
                    idx++;
                    todo &= ~m;
                }
                if (valid) { REL[r] = (uint16_t)myrel; IDX[r] = (uint8_t)myidx; }
                if (lane == 0) NG[blk] = (uint8_t)idx;            // <= 64
            }
            __syncthreads();
            if (threadIdx.x < 64) {                                                // (B)
                for (u32 blk = 0; blk < nblk; blk++) {
                    if (lane < (u32)NG[blk]) {
                        const u32 g0 = LG[blk * 64u + lane], start = lds.run[g0];
                        lds.run[g0] = start + LT[blk * 64u + lane];
                        LT[blk * 64u + lane] = start;
                    }
                }
            }
            __syncthreads();
            for (u32 blk = wv; blk < nblk; blk += GT / 64) {                       // (C)
                const u32 r = blk * 64u + lane;
                if (r < S) {
                    const u32 tl = ndigits(r + 1) + 1;
                    store_token(seds + LT[blk * 64u + IDX[r]] + REL[r], token_of(r, tl), tl);
                }
            }
            __builtin_amdgcn_s_waitcnt(0);                     // every wave's stores have landed before the braces overwrite the last ','
            __syncthreads();
            if (threadIdx.x < 64)
                for (u32 g = lane; g < k; g += 64) seds[lds.run[g] - 1] = '}';
        } else if (threadIdx.x < 64) {
            for (u32 base = 0; base < S; base += 64) {
                const u32 r = base + lane;
                const bool valid = r < S;
                const u32 g = valid ? lds.gid[r] : 0xffffffffu;
                const u32 tl = ndigits(r + 1) + 1;
                const u32 tlA = __shfl(tl, 0, 64);              // <= 2 token lengths per 64 rows
                const u64 maskA = ballot64(valid && tl == tlA);
                // placement: one wave-uniform step per distinct group of the block; the stores follow the
                // loop, all lanes together (inside it they would run once per group, a few lanes at a time)
                u32 myoff = 0;
                u64 todo = ballot64(valid);
                while (todo) {
                    const int leader = __builtin_ctzll(todo);
                    const u32 g0 = (u32)__builtin_amdgcn_readlane((int)g, leader);
                    const u64 m = ballot64(valid && g == g0);
                    const u32 start = lds.run[g0];
                    if (valid && g == g0) myoff = start + mbcnt(m & maskA) * tlA + mbcnt(m & ~maskA) * (tlA + 1);
                    const u32 tot = __builtin_popcountll(m & maskA) * tlA +
                                    __builtin_popcountll(m & ~maskA) * (tlA + 1);
                    if (lane == (u32)leader) lds.run[g0] = start + tot;
                    todo &= ~m;
                }
                if (valid) store_token(seds + myoff, token_of(r, tl), tl);
            }
            // every store above must have landed before the closing braces overwrite the last ','
            __builtin_amdgcn_s_waitcnt(0);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            for (u32 g = lane; g < k; g += 64) seds[lds.run[g] - 1] = '}';
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// Fast path (S <= 1024): one WAVE per variant segment, rows in registers.
//   vc columns are in natural row order; lane l owns rows 16l .. 16l+15 (one 16-byte load at column + 16l,
//   byte i = row 16l+i), so row order = (lane, byte) and a lane's ids are 16 consecutive numbers.
//   Per-row state is SWAR bytes in a uint4: gid (group id), rm (rows not yet grouped).
//   Grouping (k_seg_group): rows are grouped by RAW equality over the segment's columns ('-' and '\n'
//   normalised), which needs no per-row gap stripping; the gap-stripped string of each raw group is then
//   built once from its first row, and raw groups spelling the same string are joined.  Every path is exact
//   (msa_transforms.cpp:262-293); what the wave cannot decide exactly (NUL bytes, long strings whose hashed
//   keys meet, more than KCAP strings) goes to the generic workgroup-per-segment kernels.
//   The result is a grouping RECORD per segment: group id of every row (2 bits when there are at most 4
//   strings, else 4 bits; natural row order, lane l's rows in dword(s) l), number of strings, first rows.
//   Text (k_emit_fast): ids of rows 0..127 are placed one row per lane (mixed token lengths), ids 129.. by
//   the lane that owns the 16 rows: per-lane cursors in LDS, one ds_add_rtn + one aligned ds_write_b32 per id.
// ---------------------------------------------------------------------------------------------
constexpr u64 META_REC = 1ull << 63;      // the segment has a grouping record; low 40 bits = record index
constexpr u64 META_KIND4 = 1ull << 62;    // 4-bit group ids (5..16 strings), else 2-bit
constexpr u64 META_INLINE = 1ull << 61;   // the .eds text of the segment is in the record
constexpr u64 META_KIND8 = 1ull << 60;    // 8-bit group ids (17..64 strings)
constexpr u64 META_RECID = (1ull << 40) - 1;
constexpr u64 CNT_SCATTER = 1ull << 63;   // count-list descriptor: ncol << 48 | slot of the first column
constexpr u64 CNT_SLOT = (1ull << 48) - 1;
constexpr u64 CNT_MIXED = 1ull << 62;     // a segment of an l-EDS with common columns between its variant runs (heavy grouping kernel)
// record header (behind the group ids): +0 u32 k | textlen << 8 | ncol << 16;  +8 u64 slot0 | CNT_SCATTER;
// +16 u16 rep[16] (first row of every string);  +48 text[80]
constexpr u32 REC_H_SLOT = 8, REC_H_REP = 16;

// thread per segment: sizes of common segments and of the variant segments the column scan grouped itself; a
// variant segment of pure variant columns goes on the work list of the wave-per-segment grouping kernel (its
// ordinal vi among the variant segments = its record index, and its column descriptor), the others go to the
// generic kernels
__global__ void __launch_bounds__(256) k_seg_meta(FastParams p)
{
    const MsaView& mv = p.mv;
    if (mv.hdr->status) return;                           // vc overflow: slots past the capacity exist; the host grows vc and replans
    const u64 nseg = *p.nseg_ptr;
    const u64 p0 = mv.vbit(0) ? 0 : 1;                   // variant and common segments alternate
    const u64 nvs = nseg > p0 ? (nseg - p0 + 1) / 2 : 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) mv.hdr->nvs = nvs;
    auto common = [&](u64 seg, u64 a, u64 b) {
        p.eds_len[seg] = 2 + (b - a);
        p.seds_len[seg] = 3;
        p.segmeta[seg] = 0;
        if (common_is_long(b - a)) p.long_list[atomicAdd(p.long_count, 1ull)] = seg;
    };
    // A thread per VARIANT segment: it also does the common segment behind it (and the one in front of the first), so
    // every lane of a wave walks the same chain of dependent loads (segment start -> slot -> record info) instead of
    // every other lane idling through it.
    if (nvs == 0) {                                      // a single common segment
        if (blockIdx.x == 0 && threadIdx.x == 0 && nseg) common(0, p.seg_start[0], p.seg_start[1]);
        return;
    }
    for (u64 vi = blockIdx.x * (u64)blockDim.x + threadIdx.x; vi < nvs; vi += (u64)gridDim.x * blockDim.x) {
        const u64 seg = 2 * vi + p0;
        const u64 a = p.seg_start[seg], b = p.seg_start[seg + 1];
        if (vi == 0 && p0) common(0, p.seg_start[0], a);
        if (seg + 1 < nseg) common(seg + 1, b, p.seg_start[seg + 2]);
        u64 work_cm = 0, work_wide = 0, work_wide16 = 0;
        bool done = false;
        if (p.Fraw && ((p.Fraw[a >> 6] >> (a & 63)) & 1ull)) {           // a run the column scan grouped itself?
            const u64 slot = mv.slot(a);
            const u32 info = p.rec_info[slot];
            if (info >> 31) {
                p.eds_len[seg] = (info >> 8) & 0xffu;
                p.seds_len[seg] = (u64)(info & 0xffu) + p.tok_total;
                p.segmeta[seg] = META_REC | META_INLINE | ((info >> 30) & 1u ? META_KIND4 : 0) | slot;
                work_wide = ((info >> 30) & 1u) && (info & 0xffu) <= 8u;            // 5..8 strings / 9..16: the two wide lists
                work_wide16 = ((info >> 30) & 1u) && (info & 0xffu) > 8u;
                done = true;
            }
        }
        if (!done) {
            u64 cm = 0;                                       // 0: generic kernels
            const u64 ncol = b - a;
            if (ncol <= 64) {
                const u64 s0 = mv.slot(a);
                bool pure = true, contig = true;
                for (u64 c = a + 1; c < b; c++) {
                    pure = pure && mv.vbit(c);
                    if (pure) contig = contig && mv.slot(c) == s0 + (c - a);
                }
                if (pure) cm = (ncol << 48) | s0 | (contig ? 0 : CNT_SCATTER);
                else cm = (ncol << 48) | s0 | CNT_MIXED;      // context merge: common columns inside (every row has the reference byte there)
            }
            if (!cm) p.slow_list[atomicAdd(p.slow_count, 1ull)] = seg;        // too wide or mixed columns
            p.segmeta[seg] = 0;                               // k_seg_group fills it in
            work_cm = cm;
        }
        // light list: up to ten pure variant columns; heavy list: 11..64 columns or common columns inside
        const bool heavy = work_cm && ((work_cm & CNT_MIXED) || ((work_cm >> 48) & 0xffu) > 10u);
        p.cnt_meta[vi] = work_cm; p.cnt_flag[vi] = work_cm && !heavy ? 1 : 0; p.heavy_flag[vi] = heavy ? 1 : 0;
        p.wide_flag[vi] = work_wide; p.wide16_flag[vi] = work_wide16;
    }
}

// work list of the grouping kernel: the variant segments with a column descriptor, compacted with a scan of the
// flags (no atomics: 4 M appends to one counter would take milliseconds)
__global__ void __launch_bounds__(256) k_work_scatter(FastParams p, const u64* __restrict__ pos, const u64* __restrict__ hpos,
                                                      const u64* __restrict__ nvs_ptr)
{
    if (p.mv.hdr->status) return;
    const u64 nvs = *nvs_ptr;
    for (u64 vi = blockIdx.x * (u64)blockDim.x + threadIdx.x; vi < nvs; vi += (u64)gridDim.x * blockDim.x) {
        const u64 cm = p.cnt_meta[vi];
        if (cm) {
            const bool heavy = (cm & CNT_MIXED) || ((cm >> 48) & 0xffu) > 10u;
            if (heavy) { const u64 i = hpos[vi]; p.heavy_vi[i] = vi; p.heavy_cm[i] = cm; }
            else { const u64 i = pos[vi]; p.cnt_vi[i] = vi; p.cnt_cm[i] = cm; }
        }
        const u64 wpos = p.wide_flag[vi];               // (exclusive scan in place: position; a set flag = the next one is larger)
        const u64 wnext = vi + 1 < nvs ? p.wide_flag[vi + 1] : *p.wide_count;
        if (wnext != wpos) p.wide_list[wpos] = vi;
        const u64 xpos = p.wide16_flag[vi];
        const u64 xnext = vi + 1 < nvs ? p.wide16_flag[vi + 1] : *p.wide16_count;
        if (xnext != xpos) p.wide16_list[xpos] = vi;
    }
}

// four lookups in a table of 32 bytes (t[2q+1]:t[2q] holds entries 8q .. 8q+7): byte i of the result = table[byte i of g4]
__device__ __forceinline__ uint32_t lut32(const u32* t, uint32_t g4)
{
    const uint32_t sel = g4 & 0x07070707u;
    const uint32_t h3 = (g4 >> 3) & 0x01010101u, m3 = (h3 << 8) - h3;      // 0xFF where bit 3 of the index is set
    const uint32_t h4 = (g4 >> 4) & 0x01010101u, m4 = (h4 << 8) - h4;      // ... bit 4
    const uint32_t e0 = __builtin_amdgcn_perm(t[1], t[0], sel), e1 = __builtin_amdgcn_perm(t[3], t[2], sel);
    const uint32_t e2 = __builtin_amdgcn_perm(t[5], t[4], sel), e3 = __builtin_amdgcn_perm(t[7], t[6], sel);
    const uint32_t lo = (e1 & m3) | (e0 & ~m3), hi = (e3 & m3) | (e2 & ~m3);
    return (hi & m4) | (lo & ~m4);
}

// ... of 64 bytes
__device__ __forceinline__ uint32_t lut64(const u32* t, uint32_t g4)
{
    const uint32_t h5 = (g4 >> 5) & 0x01010101u, m5 = (h5 << 8) - h5;
    return (lut32(t + 8, g4) & m5) | (lut32(t, g4) & ~m5);
}

template <int MAXG> __device__ __forceinline__ uint32_t lutN(const u32* t, uint32_t g4)
{
    if constexpr (MAXG <= 8) return __builtin_amdgcn_perm(t[1], t[0], g4);
    else if constexpr (MAXG <= 16) {
        const uint32_t sel = g4 & 0x07070707u, h = (g4 >> 3) & 0x01010101u, m = (h << 8) - h;   // 0xFF where the index is >= 8
        return (__builtin_amdgcn_perm(t[3], t[2], sel) & m) | (__builtin_amdgcn_perm(t[1], t[0], sel) & ~m);
    } else if constexpr (MAXG <= 32) return lut32(t, g4);
    else return lut64(t, g4);
}

// Any alphabet, up to 64 columns: refine the partition of the rows column by column.  All rows start in one
// raw group; per column every row is compared with the byte of its group's first row (one table lookup per row:
// v_perm_b32 on the group id), and a group whose rows disagree is split off at its first disagreeing row.  Raw
// groups are classes of identical rows, exact for every byte value; their gap-stripped strings (the reference
// ends a row's string at NUL, msa_transforms.cpp:282) are built once from the first rows, and raw groups that
// spell one string are joined.  More than 64 raw groups, or a NUL inside the segment: generic kernels.
// Returns 1 grouped, 0 generic kernels (NUL), -1 more than MAXG raw groups.
// load_col(c): this lane's 16 bytes of column c; cell(c, row): one byte (a common column inside an l-EDS segment is the
// reference byte in every row: it splits no group, but its letter belongs to every string).
template <int MAXG, bool CHECK_NL, class LoadCol, class Cell>
__device__ __forceinline__ int refine_groups(LoadCol load_col, Cell cell, u32 ncol, const uint4& col0, u32 lane, const uint4& vmask,
                                             FastGroups& G, u32& saw_nl, uint8_t* strs)
{
    uint4 gid = make_uint4(~vmask.x, ~vmask.y, ~vmask.z, ~vmask.w);        // group 0; rows that do not exist: 0xFF
    u32 k = 1, rep_l = 0;                               // lane g: first row of raw group g
    for (u32 c0 = 0; c0 < ncol; c0 += 4) {
        uint4 cvs[4];                                   // four column loads in flight
#pragma unroll
        for (int j = 0; j < 4; j++) {
            cvs[j] = make_uint4(0, 0, 0, 0);
            if (c0 + j < ncol) cvs[j] = (c0 + j == 0) ? col0 : load_col(c0 + j);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (c0 + j >= ncol) continue;
            if (cell(c0 + j, ~0u) < 0x100u) continue;           // a common column splits no group (row ~0: "is it common?")
            const uint4 col = cvs[j];
            u32 t[MAXG / 4];                                  // byte g & 7 of (t[2q+1]:t[2q]), q = g >> 3: this column's byte of group g's first row
#pragma unroll
            for (int i = 0; i < MAXG / 4; i++) t[i] = 0;
            for (u32 g = 0; g < k; g++) {
                const u32 r = (u32)__builtin_amdgcn_readlane((int)rep_l, (int)g);
                const u32 tb = leader_byte(col, (int)(r >> 4), r & 15u) << ((g & 3u) * 8u);
#pragma unroll
                for (int i = 0; i < MAXG / 4; i++) if ((g >> 2) == (u32)i) t[i] |= tb;
            }
            for (;;) {
                uint4 ex;
                if (k <= 8u) {
                    ex = make_uint4(__builtin_amdgcn_perm(t[1], t[0], gid.x), __builtin_amdgcn_perm(t[1], t[0], gid.y),
                                    __builtin_amdgcn_perm(t[1], t[0], gid.z), __builtin_amdgcn_perm(t[1], t[0], gid.w));
                } else ex = make_uint4(lutN<MAXG>(t, gid.x), lutN<MAXG>(t, gid.y), lutN<MAXG>(t, gid.z), lutN<MAXG>(t, gid.w));
                const uint4 mm = make_uint4((ex.x ^ col.x) & vmask.x, (ex.y ^ col.y) & vmask.y, (ex.z ^ col.z) & vmask.z, (ex.w ^ col.w) & vmask.w);
                const u64 B = ballot64(any4(mm));
                if (!B) break;
                if (k >= (u32)MAXG) return -1;
                // the first row (in row order) that disagrees with its group's first row starts a new group: the rows of
                // its old group that have its byte in this column
                const int ld = __builtin_ctzll(B);
                const uint4 nzm = make_uint4(bytes_ne_mask(mm.x, 0u), bytes_ne_mask(mm.y, 0u), bytes_ne_mask(mm.z, 0u), bytes_ne_mask(mm.w, 0u));
                const u32 ix = (u32)__builtin_amdgcn_readlane((int)first_byte_index(nzm), ld);
                const u32 gold = leader_byte(gid, ld, ix), bnew = leader_byte(col, ld, ix);
                const uint4 e1 = bytes_eq_mask(gid, gold * 0x01010101u), e2 = bytes_eq_mask(col, bnew * 0x01010101u);
                const uint4 em = make_uint4(e1.x & e2.x, e1.y & e2.y, e1.z & e2.z, e1.w & e2.w);
                const uint32_t kk = k * 0x01010101u;
                gid.x = (gid.x & ~em.x) | (em.x & kk); gid.y = (gid.y & ~em.y) | (em.y & kk);
                gid.z = (gid.z & ~em.z) | (em.z & kk); gid.w = (gid.w & ~em.w) | (em.w & kk);
                if (lane == k) rep_l = (u32)ld * 16u + ix;
#pragma unroll
                for (int i = 0; i < MAXG / 4; i++) if ((k >> 2) == (u32)i) t[i] |= bnew << ((k & 3u) * 8u);
                k++;
            }
        }
    }
    // ---- raw groups in the order of their first rows; strings of their first rows: lane = column
    u32 rank_l = 0;
    for (u32 g = 0; g < k; g++) rank_l += (u32)__builtin_amdgcn_readlane((int)rep_l, (int)g) < rep_l ? 1u : 0u;
    // cells of the first rows -> LDS (raw[r][column], r-th raw group in first-row order), eight rows' loads in flight
    uint8_t* raw = strs + 4096;
    for (u32 r0 = 0; r0 < k; r0 += 8) {
        u32 chv[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            chv[i] = 0;
            if (r0 + i < k) {
                const u32 gs = (u32)__builtin_ctzll(ballot64(lane < k && rank_l == r0 + i));
                const u32 row = (u32)__builtin_amdgcn_readlane((int)rep_l, (int)gs);
                if (lane < ncol) {
                    chv[i] = cell(lane, row);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 8; i++) if (r0 + i < k) raw[(r0 + i) * 64u + lane] = (uint8_t)chv[i];   // (lanes >= ncol: 0)
    }
    u32 lut[MAXG / 4];                                  // raw group -> final group
#pragma unroll
    for (int i = 0; i < MAXG / 4; i++) lut[i] = 0;
    u32 nroot = 0, sumlen = 0, root_len_l = 0, root_slot_l = 0, root_rep_l = 0, root_hash_l = 0;   // lane t: root t
    for (u32 r = 0; r < k; r++) {
        const u32 gs = (u32)__builtin_ctzll(ballot64(lane < k && rank_l == r));
        const u32 row = (u32)__builtin_amdgcn_readlane((int)rep_l, (int)gs);
        const u32 c = raw[r * 64u + lane];
        const u64 nulb = ballot64(c == 0u);             // lanes >= ncol hold 0
        if (ncol < 64u ? (nulb & ((1ull << ncol) - 1ull)) != 0 : nulb != 0) return 0;   // a NUL ends a row's string (:282): generic kernels
        const bool valid = lane < ncol;
        if (CHECK_NL && lane < ncol && c == '\n') saw_nl = 1;
        const bool keep = valid && c != '-' && c != '\n';
        const u64 nz = ballot64(keep);
        const u32 len = (u32)__builtin_popcountll(nz);
        if (keep) strs[r * 64u + mbcnt(nz)] = (uint8_t)c;
        // the same string as an earlier root?  Candidates by length and a hash of the letters in their columns (one
        // ballot instead of a walk over the roots: wide segments of an l-EDS have tens of raw groups), then the letters.
        const u32 hsh = wave_xor_all(keep ? (c + 1u) * (0x9e3779b1u + 0x85ebca77u * mbcnt(nz)) : 0u);
        u32 fin = nroot;
        for (u64 cand = ballot64(lane < nroot && root_len_l == len && root_hash_l == hsh); cand; cand &= cand - 1) {
            const u32 tt = (u32)__builtin_ctzll(cand);
            const u32 sl = (u32)__builtin_amdgcn_readlane((int)root_slot_l, (int)tt);
            const bool diff = lane < len && strs[r * 64u + lane] != strs[sl * 64u + lane];
            if (!ballot64(diff)) { fin = tt; break; }
        }
        if (fin == nroot) {
            if (lane == nroot) { root_len_l = len; root_slot_l = r; root_rep_l = row; root_hash_l = hsh; }
            nroot++; sumlen += len;
        }
        // lut[gs >> 2] |= fin << ...  (static indices only: registers)
#pragma unroll
        for (int i = 0; i < MAXG / 4; i++) if ((gs >> 2) == (u32)i) lut[i] |= fin << ((gs & 3u) * 8u);
    }
    G.gid = make_uint4(lutN<MAXG>(lut, gid.x) | ~vmask.x, lutN<MAXG>(lut, gid.y) | ~vmask.y, lutN<MAXG>(lut, gid.z) | ~vmask.z, lutN<MAXG>(lut, gid.w) | ~vmask.w);
    G.k = nroot; G.sumlen = sumlen; G.rep = root_rep_l; G.len = root_len_l; G.key_lo = 0; G.key_hi = 0;
    return 1;
}

// Group the rows of a fast segment (msa_transforms.cpp:262-293: distinct gap-stripped strings in
// order of first appearance).  Returns false when the segment must take the generic path.
//   one column : DNA table lookups, else exact SWAR byte compares.
//   2..20 cols over {A,C,G,T,N,-}: exact 3-bit-per-column keys.
//   otherwise  : every row gets a 96-bit additive signature of its raw column bytes (gaps normalised
//                to 0; v_mad_u32_u24 = full rate); rows with equal signatures are PROPOSED as a raw
//                group and then compared with the group's first row byte for byte (phase B), so the
//                grouping is exact; raw groups that spell the same string are joined by their
//                stripped string (verbatim key up to 12 letters; longer strings that hash alike send
//                the segment to the generic kernels).  NUL bytes (msa_transforms.cpp:282) -> generic.
// HEAVY false: one column, or 2..10 columns over the DNA alphabet; everything else returns 2 (= try the heavy
// instantiation, which needs twice the registers).  1 = grouped, 0 = generic kernels.
template <bool CHECK_NL, bool HEAVY>
__device__ __forceinline__ int fast_group(const MsaView& mv, u64 seg_a, u64 cmeta, const uint4& col0, u32 rb, u32 lane,
                                          const uint4& vmask, FastGroups& G, u32& saw_nl, uint8_t* strs /* HEAVY: 2 x 64 x 64 bytes of LDS */)
{
    const u32 ncol = (u32)(cmeta >> 48) & 0xffu;
    uint4 rm = vmask;
    G.gid = make_uint4(~0u, ~0u, ~0u, ~0u);
    G.k = 0; G.sumlen = 0; G.key_lo = 0; G.key_hi = 0; G.rep = 0; G.len = 0;
    int leader;
    u32 i0;

    if (ncol == 1) {
        if (fast_group_dna1(col0, rb, vmask, lane, mv.S, G)) return 1;
        G.gid = make_uint4(~0u, ~0u, ~0u, ~0u);
        G.k = 0; G.sumlen = 0; G.key_lo = 0; G.key_hi = 0; G.rep = 0; G.len = 0;
        // a NUL byte ends the row's string in the reference (msa_transforms.cpp:282): exact kernels only
        if (ballot64(any_nul(col0, vmask))) return 0;
        const uint4 col = normalise_col<CHECK_NL>(col0, vmask, saw_nl);
        while (first_remaining(rm, leader, i0)) {
            const u32 c = leader_byte(col, leader, i0);
            uint4 eq = bytes_eq_mask(col, c * 0x01010101u);
            eq.x &= rm.x; eq.y &= rm.y; eq.z &= rm.z; eq.w &= rm.w;
            if (!fast_assign(G, rm, eq, (u64)c, 0ull, c ? 1u : 0u, lane, (u32)leader * 16u + i0)) return 0;
        }
        return 1;
    }

    const u64 slot0 = cmeta & CNT_SLOT;
    const bool scatter = (cmeta & CNT_SCATTER) != 0, mixed = (cmeta & CNT_MIXED) != 0;
    if (mixed && !HEAVY) return 2;
    const u32 loff = lane * 16u < mv.Spad - 16u ? lane * 16u : mv.Spad - 16u;     // lanes without rows stay inside the column
    const uint8_t* cbase = mv.vc + slot0 * (u64)mv.Spad + loff;
    // A segment whose slots are not consecutive crosses a tile edge of the column scan (once: it has at most 64
    // columns): the columns from the edge on have the slots word_slot[edge / 64] + 0, 1, ..  (one-line rows)
    u32 nA = ncol;
    const uint8_t* cbaseB = cbase;
    const bool two = scatter && mv.lw == 0 && mv.tileW != 0;
    if (two) {
        const u64 bnd = (seg_a / mv.tileW + 1) * mv.tileW;
        nA = (u32)(bnd - seg_a);
        cbaseB = mv.vc + uniform64(mv.word_slot[bnd >> 6]) * (u64)mv.Spad + loff;
    }
    auto col_ptr = [&](u32 c) -> const uint8_t* {
        if (two) return c < nA ? cbase + (u64)c * mv.Spad : cbaseB + (u64)(c - nA) * mv.Spad;
        return scatter ? mv.vc + mv.slot(seg_a + c) * (u64)mv.Spad + loff : cbase + (u64)c * mv.Spad;
    };

    if (!HEAVY && ncol > 10u) return 2;
    if (ncol <= 10u && !mixed) {
        auto load_col = [&](u32 c) -> uint4 { return load16u(col_ptr(c)); };
        const int r = fast_group_dnakeys<1>(load_col, ncol, col0, lane, vmask, G);
        if (r) return r > 0 ? 1 : 0;
        G.gid = make_uint4(~0u, ~0u, ~0u, ~0u);          // another alphabet
        G.k = 0; G.sumlen = 0; G.key_lo = 0; G.key_hi = 0; G.rep = 0; G.len = 0;
    }
    if constexpr (HEAVY) {
        if (ncol > 10u && ncol <= 20u && !mixed) {         // 11..20 columns over the DNA alphabet: exact keys in two dwords
            // (fetching the columns into LDS by LDS-DMA in one round trip - no registers for more than four loads in flight -
            // was built and measured in round 3: 1.33 vs 1.35 ms, the kernel is bound by its instructions, not by the loads)
            auto load_col = [&](u32 c) -> uint4 { return load16u(col_ptr(c)); };
            const int r = fast_group_dnakeys<2>(load_col, ncol, col0, lane, vmask, G);
            if (r) return r > 0 ? 1 : 0;
            G.gid = make_uint4(~0u, ~0u, ~0u, ~0u);          // another alphabet
            G.k = 0; G.sumlen = 0; G.key_lo = 0; G.key_hi = 0; G.rep = 0; G.len = 0;
        }
        auto cell_ptr = [&](u32 c) -> const uint8_t* {     // column c, row 0
            const uint8_t* cp = two ? (c < nA ? cbase + (u64)c * mv.Spad : cbaseB + (u64)(c - nA) * mv.Spad)
                                    : (scatter ? mv.vc + mv.slot(seg_a + c) * (u64)mv.Spad + loff : cbase + (u64)c * mv.Spad);
            return cp - loff;
        };
        // mixed segment: lane c looks its column up once (variant: slot in vc; common: the reference byte) - one round of
        // dependent loads for all columns together instead of one per column inside the refinement loop
        u32 cs_lo = 0, cs_hi = 0, cref = 0x100u;               // cref < 0x100: a common column
        if (mixed && lane < ncol) {
            if (mv.vbit(seg_a + lane)) { const u64 sl = mv.slot(seg_a + lane); cs_lo = (u32)sl; cs_hi = (u32)(sl >> 32); }
            else cref = mv.ref_byte(seg_a + lane);
        }
        auto load_colx = [&](u32 c) -> uint4 {                 // c is wave-uniform
            if (mixed) {
                const u32 rb = (u32)__builtin_amdgcn_readlane((int)cref, (int)c);
                if (rb < 0x100u) { const u32 b = rb * 0x01010101u; return make_uint4(b, b, b, b); }
                const u64 sl = ((u64)(u32)__builtin_amdgcn_readlane((int)cs_hi, (int)c) << 32) | (u32)__builtin_amdgcn_readlane((int)cs_lo, (int)c);
                return load16u(mv.vc + sl * (u64)mv.Spad + loff);
            }
            return load16u(col_ptr(c));
        };
        auto cell = [&](u32 c, u32 row) -> u32 {               // row == ~0: (uniform c) the reference byte of a common column, else 0x100
            if (row == ~0u) return mixed ? (u32)__builtin_amdgcn_readlane((int)cref, (int)c) : 0x100u;
            if (mixed) return cref < 0x100u ? cref : (u32)mv.vc[(((u64)cs_hi << 32) | cs_lo) * (u64)mv.Spad + row];   // c == lane
            return (u32)cell_ptr(c)[row];
        };
        int r = refine_groups<16, CHECK_NL>(load_colx, cell, ncol, col0, lane, vmask, G, saw_nl, strs);
        if (r < 0) r = refine_groups<64, CHECK_NL>(load_colx, cell, ncol, col0, lane, vmask, G, saw_nl, strs);   // (rare: 17..64 raw groups)
        return r > 0 ? 1 : 0;
    } else return 2;
}

// K3 fast: grouping records + sizes of the variant segments on the work list, one wave per segment.  The light
// instantiation (one column / up to ten DNA columns) hands what it cannot do to the heavy one's list.
// The descriptor and the first column of the wave's next segment are requested one iteration ahead.
template <bool HEAVY>
__global__ void __launch_bounds__(256, HEAVY ? 3 : 4) k_seg_group(FastParams p, const u64* __restrict__ lvi, const u64* __restrict__ lcm,
                                                                  const u64* __restrict__ n_ptr)
{
    __shared__ uint8_t strs_all[HEAVY ? 4 * 8192 : 4];
    uint8_t* strs = strs_all + (HEAVY ? (threadIdx.x >> 6) * 8192u : 0u);
    const MsaView& mv = p.mv;
    if (mv.hdr->status) return;
    const u32 lane = threadIdx.x & 63;
    const u64 p0 = mv.vbit(0) ? 0 : 1;
    const u64 n = *n_ptr;
    const u64 nw = ((u64)gridDim.x * blockDim.x) >> 6;
    const uint4 vmask = fast_valid_mask(lane, mv.S);
    const u32 nl = (mv.S + 15u) >> 4;                          // lanes that own rows
    const u32 loff = lane * 16u < mv.Spad - 16u ? lane * 16u : mv.Spad - 16u;
    u32 saw_nl = 0;
    auto load_col = [&](u64 cm) -> uint4 {
        return cm ? load16u(mv.vc + (cm & CNT_SLOT) * (u64)mv.Spad + loff) : make_uint4(0, 0, 0, 0);
    };
    auto load_rb = [&](u64 cm) -> u32 {                        // the first column's byte of row `lane`
        return cm && lane < mv.S ? (u32)mv.vc[(cm & CNT_SLOT) * (u64)mv.Spad + lane] : 0u;
    };
    u64 it = (u64)blockIdx.x * (blockDim.x >> 6) + uniform32(threadIdx.x >> 6);
    u64 cmeta = it < n ? uniform64(lcm[it]) : 0, vi = it < n ? uniform64(lvi[it]) : 0;
    u64 cmeta_n = it + nw < n ? uniform64(lcm[it + nw]) : 0, vi_n = it + nw < n ? uniform64(lvi[it + nw]) : 0;
    uint4 col = load_col(cmeta);
    u32 rb = load_rb(cmeta);
    while (it < n) {
        const u64 seg = 2 * vi + p0;
        // prefetch: next segment's first column and the descriptor after it
        const uint4 col_n = load_col(cmeta_n);
        const u32 rb_n = load_rb(cmeta_n);
        const u64 i2 = it + 2 * nw < n ? it + 2 * nw : it;
        const u64 cm_v = lcm[i2], vi_v = lvi[i2];             // scalar after the wait below
        const u32 ncol = (u32)(cmeta >> 48) & 0xffu;
        FastGroups G;
        const int ok = fast_group<true, HEAVY>(mv, (cmeta & (CNT_SCATTER | CNT_MIXED)) ? uniform64(p.seg_start[seg]) : 0, cmeta, col, rb, lane, vmask, G, saw_nl, strs);
        // wait for the prefetched column here, before this segment's stores are queued behind it
        // (vmcnt retires in issue order)
        asm volatile("" :: "v"(col_n.x), "v"(col_n.y), "v"(col_n.z), "v"(col_n.w), "v"(rb_n), "v"(cm_v), "v"(vi_v));
        if (ok == 1) {
            uint8_t* rec = p.rec + vi * (u64)p.rec_stride;
            if (lane < nl) {
                if (G.k <= 4u) *reinterpret_cast<u32*>(rec + lane * 4u) = pack_gid2(G.gid, vmask);
                else if (G.k <= 16u) *reinterpret_cast<uint2*>(rec + lane * 8u) = pack_gid4(G.gid, vmask);
                else *reinterpret_cast<uint4*>(rec + lane * 16u) = G.gid;
            }
            uint8_t* hdr = rec + p.rec_gid;
            if (lane < G.k) *reinterpret_cast<uint16_t*>(hdr + REC_H_REP + lane * 2u) = (uint16_t)G.rep;
            if (lane == 0) {
                *reinterpret_cast<u32*>(hdr) = G.k | (ncol << 16);
                *reinterpret_cast<u64*>(hdr + REC_H_SLOT) = cmeta & (CNT_SLOT | CNT_SCATTER | CNT_MIXED);
                p.eds_len[seg] = 2 + (u64)(G.k - 1) + G.sumlen;
                p.seds_len[seg] = (u64)G.k + p.tok_total;
                p.segmeta[seg] = META_REC | (G.k > 16u ? META_KIND8 : G.k > 4u ? META_KIND4 : 0) | vi;
                if (G.k > 8u) p.wide16_list[atomicAdd(p.wide16_count, 1ull)] = vi;
                else if (G.k > 4u) p.wide_list[atomicAdd(p.wide_count, 1ull)] = vi;
            }
        } else if (lane == 0 && !(!HEAVY && ok == 2)) {
            p.slow_list2[atomicAdd(p.slow_count2, 1ull)] = seg;       // the generic kernels take it
        }
        if (!HEAVY && ok == 2 && lane == 0) {                       // another alphabet: the second heavy pass takes it
            const u64 i = atomicAdd(p.heavy2_n, 1ull);
            p.heavy2_vi[i] = vi; p.heavy2_cm[i] = cmeta;
        }
        const u64 cmeta_nn = it + 2 * nw < n ? uniform64(cm_v) : 0, vi_nn = it + 2 * nw < n ? uniform64(vi_v) : 0;
        it += nw; cmeta = cmeta_n; cmeta_n = cmeta_nn; vi = vi_n; vi_n = vi_nn; col = col_n; rb = rb_n;
    }
    if (saw_nl) atomicOr(&mv.hdr->status, (u64)(ST_LAYOUT | ST_NEWLINE_IN_DATA));
}

// ---- K5 fast: .seds / .eds text of the variant segments.  msa_transforms.cpp:297-317.
constexpr int EM_STAGE = 4608;     // >= tokens of 1024 rows (4013) + 16 braces + 6 bytes of padding per string
constexpr int EM_TRASH = 256;      // one dword per lane behind it: where the tokens of rows that are not placed go
template <int ROWS> struct EmitWaveLdsT {
    alignas(16) uint8_t stage[EM_STAGE + EM_TRASH];   // the segment's id lists, one 4-aligned region per string
    alignas(16) u32 tab[ROWS * 64];        // [string][lane] (bank = lane: conflict-free): where this lane's next id of that
                                           // string goes (byte offset in stage); last row (4 / 16): dummies
    alignas(16) u32 gt[64];                // per string: [0..15] start of its ids, [16..31] start of the ids >= 129,
};                                         // [32..47] / [48..63] LDS source / global destination of its full 16-byte chunks

// Token bytes -> LDS as single-byte stores.  Written in C (d[0] = ..; d[1] = ..) hipcc fuses the stores into one
// ds_write_b32 at an unaligned address, which the LDS executes ~10x slower.  a = LDS byte address.
__device__ __forceinline__ void lds_put2(u32 a, u32 t)
{
    const u32 t8 = t >> 8;
    asm volatile("ds_write_b8 %0, %1\n\tds_write_b8 %0, %2 offset:1" :: "v"(a), "v"(t), "v"(t8) : "memory");
}
template <int OFF> __device__ __forceinline__ void lds_put1(u32 a, u32 v)
{
    asm volatile("ds_write_b8 %0, %1 offset:%2" :: "v"(a), "v"(v), "n"(OFF) : "memory");
}
__device__ __forceinline__ u32 lane_read(u32 v, u32 src_lane)       // v of lane src_lane (per-lane source)
{
    return (u32)__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), (int)v);
}

// Writes the id lists "{i,i,..}{i,..}.." of k (<= 4 / <= 16) strings of one segment to gseds (msa_transforms.cpp:305-316)
// and returns their bytes.
//   BITS   2: x0 = this lane's 16 group ids (2 bits each);  4: x0 = rows 0..7, x1 = rows 8..15 (4 bits each)
//   am     bit j: row 16*lane+j (>= 128) is placed;  g0/v0, g1/v1: group id / "is placed" of rows `lane`, `64 + lane`
//   rows 0..127 (ids 1..128: 2-, 3- and 4-byte tokens) are placed one row per lane: per block of 64 rows ONE packed
//   DPP wave scan ranks the rows of four strings at once (the token lengths added into 8-bit fields).
//   rows 128.. (4-byte tokens up to id 999): the owning lane walks its 16 rows; its cursor of every string lives
//   in LDS (tab): 16 returning ds_add hand out the positions and advance the cursors, then 16 aligned ds_write_b32
//   place the tokens (rows that are not placed use a dummy cursor: no branches, all 16 atomics in flight together).
//   Every string's region of `stage` is padded so that its 4-byte tokens are dword-aligned; the regions are copied
//   out one by one (16-byte LDS reads, unaligned 16-byte global stores, byte stores for the ragged ends).
template <int BITS, bool HAS5, int KMAX, class Lds, class PreFlush>
__device__ __forceinline__ u32 emit_ids(u32 x0, u32 x1, u32 am, u32 g0, u32 g1, bool v0, bool v1, u32 k, u32 S, u32 lane,
                                        const u32 (&tokc)[16], u32 htok0, u32 htok1, Lds& L, uint8_t* gseds,
                                        PreFlush pre_flush)
{
    constexpr u32 K = BITS == 2 ? 4u : (u32)KMAX;  // table rows in use (row K: dummies); 4-bit ids: 8 or 16 strings at most
    constexpr int NQ = BITS == 2 ? 1 : KMAX / 4;   // quartets of strings
    const u32 sbase = (u32)(uintptr_t)L.stage;     // LDS byte address (low half of the flat address)
    const u32 tl0 = lane < 9u ? 2u : 3u, tl1 = lane < 35u ? 3u : 4u;       // ids 1-9 | 10-64 and 65-99 | 100-128
    u32 ex0 = 0, ex1 = 0;            // bytes of this lane's string in front of this lane's token, inside the block
    u32 tot0[NQ], tot1[NQ];          // bytes per string and block, four 8-bit fields per quartet (uniform)
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        tot0[q] = 0; tot1[q] = 0;
        if ((u32)q * 4u < k) {
            const bool in0 = v0 && (BITS == 2 || (g0 >> 2) == (u32)q), in1 = v1 && (BITS == 2 || (g1 >> 2) == (u32)q);
            const u32 f0 = in0 ? tl0 << ((g0 & 3u) * 8u) : 0u, f1 = in1 ? tl1 << ((g1 & 3u) * 8u) : 0u;
            const u32 i0 = wave_scan_incl(f0), i1 = wave_scan_incl(f1);
            if (in0) ex0 = ((i0 - f0) >> ((g0 & 3u) * 8u)) & 0xffu;
            if (in1) ex1 = ((i1 - f1) >> ((g1 & 3u) * 8u)) & 0xffu;
            tot0[q] = (u32)__builtin_amdgcn_readlane((int)i0, 63);
            tot1[q] = (u32)__builtin_amdgcn_readlane((int)i1, 63);
        }
    }
    // rows with ids >= 1000 (five bytes) among this lane's rows: bit j
    const u32 first5 = 999u > lane * 16u ? 999u - lane * 16u : 0u;
    const u32 m5 = HAS5 ? (first5 >= 16u ? 0u : (0xffffu << first5) & 0xffffu) : 0u;
    // ---- rows 128..: bytes of every string among this lane's rows, prefix over the lanes, totals
    u32 pre[K];                      // bytes of string g in the lanes before this one (rows >= 128)
    u32 bbt[BITS == 2 ? 2 : KMAX / 2];   // totals, two 16-bit fields per dword (uniform)
    if (BITS == 2) {
        // spread the row masks to the 2-bit fields
        auto spread = [](u32 m) -> u32 { m = (m | (m << 8)) & 0x00ff00ffu; m = (m | (m << 4)) & 0x0f0f0f0fu;
                                         m = (m | (m << 2)) & 0x33333333u; return (m | (m << 1)) & 0x55555555u; };
        const u32 vm = spread(am);
        const u32 lo = x0 & 0x55555555u, hi = (x0 >> 1) & 0x55555555u;
        const u32 m[4] = {vm & ~(lo | hi), lo & ~hi & vm, hi & ~lo & vm, lo & hi & vm};
        u32 c[4];
#pragma unroll
        for (int g = 0; g < 4; g++) c[g] = 4u * (u32)__builtin_popcount(m[g]);
        if (HAS5) {
            const u32 s5 = spread(m5);
#pragma unroll
            for (int g = 0; g < 4; g++) c[g] += (u32)__builtin_popcount(m[g] & s5);
        }
        const u32 p01 = c[0] | (c[1] << 16), p23 = c[2] | (c[3] << 16);
        const u32 s01 = wave_scan_incl(p01), s23 = wave_scan_incl(p23);
        const u32 e01 = s01 - p01, e23 = s23 - p23;
        pre[0] = e01 & 0xffffu; pre[1] = e01 >> 16; pre[2] = e23 & 0xffffu; pre[3] = e23 >> 16;
        bbt[0] = (u32)__builtin_amdgcn_readlane((int)s01, 63);
        bbt[1] = (u32)__builtin_amdgcn_readlane((int)s23, 63);
    } else {
#pragma unroll
        for (int g = 0; g < (int)K; g++) L.tab[g * 64 + lane] = 0;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const bool act = (am & (1u << j)) != 0;
            const u32 g = ((j < 8 ? x0 : x1) >> (4u * (j & 7))) & 15u;
            const u32 inc = act ? ((m5 >> j) & 1u ? 5u : 4u) : 0u;
            atomicAdd(&L.tab[(act ? g : K) * 64u + lane], inc);
        }
        u32 c[K];
#pragma unroll
        for (int g = 0; g < (int)K; g++) c[g] = L.tab[g * 64 + lane];
#pragma unroll
        for (int h = 0; h < (int)K / 2; h++) {
            bbt[h] = 0; pre[2 * h] = 0; pre[2 * h + 1] = 0;
            if ((u32)h * 2u < k) {
                const u32 pk = c[2 * h] | (c[2 * h + 1] << 16);
                const u32 sc = wave_scan_incl(pk), ex = sc - pk;
                pre[2 * h] = ex & 0xffffu; pre[2 * h + 1] = ex >> 16;
                bbt[h] = (u32)__builtin_amdgcn_readlane((int)sc, 63);
            }
        }
    }
    // ---- geometry of string `lane` (lanes < k): bytes of its ids up to 128 / from 129, region in `stage`, offset in the output
    u32 hb = 0, bb = 0;
    if (lane < k) {
        u32 t0 = tot0[0], t1 = tot1[0];
        if (BITS == 4) {
            const u32 q = lane >> 2;
            t0 = q == 0 ? tot0[0] : q == 1 ? tot0[NQ > 1 ? 1 : 0] : q == 2 ? tot0[NQ > 2 ? 2 : 0] : tot0[NQ > 3 ? 3 : 0];
            t1 = q == 0 ? tot1[0] : q == 1 ? tot1[NQ > 1 ? 1 : 0] : q == 2 ? tot1[NQ > 2 ? 2 : 0] : tot1[NQ > 3 ? 3 : 0];
        }
        hb = ((t0 >> ((lane & 3u) * 8u)) & 0xffu) + ((t1 >> ((lane & 3u) * 8u)) & 0xffu);
        u32 tt = bbt[0];
#pragma unroll
        for (int h = 1; h < (BITS == 2 ? 2 : (int)K / 2); h++) tt = (lane >> 1) == (u32)h ? bbt[h] : tt;
        bb = (tt >> ((lane & 1u) * 16u)) & 0xffffu;
    }
    const u32 sz = lane < k ? 1u + hb + bb : 0u;                 // '{' + tokens; the last ',' becomes '}'
    const u32 rs = lane < k ? (sz + 6u) & ~3u : 0u;              // region: up to 3 bytes of padding in front
    const u32 pk = rs | (sz << 16);
    const u32 sc = wave_scan_incl(pk), exq = sc - pk;
    const u32 total = (u32)__builtin_amdgcn_readlane((int)sc, 63) >> 16;
    const u32 P = (exq & 0xffffu) + ((3u - hb) & 3u);            // (P + 1 + hb) % 4 == 0: the 4-byte tokens are aligned
    const u32 off = exq >> 16;
    if (lane < k) { L.gt[lane] = P + 1u; L.gt[16 + lane] = P + 1u + hb; }
    // ---- cursors of this lane (rows 128..)
#pragma unroll
    for (int g = 0; g < (int)K; g++) L.tab[g * 64 + lane] = L.gt[16 + g] + pre[g];
    // ---- tokens of rows 0..127
    if (v0) {
        const u32 a = sbase + L.gt[g0] + ex0;
        lds_put2(a, htok0);
        if (lane >= 9u) lds_put1<2>(a, htok0 >> 16);
    }
    if (v1) {
        u32 t0 = tot0[0];
        if (BITS == 4) {
            const u32 q = g1 >> 2;
            t0 = q == 0 ? tot0[0] : q == 1 ? tot0[NQ > 1 ? 1 : 0] : q == 2 ? tot0[NQ > 2 ? 2 : 0] : tot0[NQ > 3 ? 3 : 0];
        }
        const u32 a = sbase + L.gt[g1] + ((t0 >> ((g1 & 3u) * 8u)) & 0xffu) + ex1;
        lds_put2(a, htok1);
        lds_put1<2>(a, htok1 >> 16);
        if (lane >= 35u) lds_put1<3>(a, htok1 >> 24);
    }
    // ---- tokens of rows 128..
    u32 at[16];
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const bool act = (am & (1u << j)) != 0;
        const u32 g = BITS == 2 ? (x0 >> (2u * j)) & 3u : ((j < 8 ? x0 : x1) >> (4u * (j & 7))) & 15u;
        const u32 inc = act ? ((m5 >> j) & 1u ? 5u : 4u) : 0u;
        at[j] = atomicAdd(&L.tab[(act ? g : K) * 64u + lane], inc);
    }
#pragma unroll
    for (int j = 0; j < 16; j++) {
        *reinterpret_cast<u32*>(L.stage + at[j]) = tokc[j];
        // a fifth byte: only rows 999.. have one (wave-uniform test: does any lane have such a row j?)
        if (HAS5 && ((j >= 7 && S > 992u + j) || S > 1008u + j)) {
            const bool five = ((am & m5) >> j) & 1u;
            L.stage[five ? at[j] + 4u : (u32)EM_STAGE + 4u * lane] = ',';
        }
    }
    asm volatile("" ::: "memory");
    // ---- braces (after the tokens: the closing one replaces the last ',')
    if (lane < k) { L.stage[P] = '{'; L.stage[P + sz - 1u] = '}'; }
    asm volatile("" ::: "memory");
    // ---- copy the regions out.  String g: LDS [P, P + sz) -> gseds + off.  hn bytes up to the first 16-byte
    // boundary of the LDS image, nfull aligned 16-byte chunks, tn bytes behind them.
    pre_flush();                                                 // (the caller's wait for its prefetched loads)
    const u32 lead = P & 15u;
    const u32 hn = lead ? (sz < 16u - lead ? sz : 16u - lead) : 0u;
    const u32 nfull = (sz - hn) >> 4;
    const u32 cinc = wave_scan_incl(nfull), cs = cinc - nfull;
    const u32 T = (u32)__builtin_amdgcn_readlane((int)cinc, 63);
    if (lane < k) { L.gt[32 + lane] = P + hn - 16u * cs; L.gt[48 + lane] = off + hn - 16u * cs; }
    for (u32 t = lane; t < T; t += 64u) {
        u32 g = 0;
        for (u32 gg = 1; gg < k; gg++) g += t >= (u32)__builtin_amdgcn_readlane((int)cs, (int)gg) ? 1u : 0u;
        const u32 src = L.gt[32 + g] + 16u * t, dst = L.gt[48 + g] + 16u * t;
        store16u(gseds + dst, *reinterpret_cast<const uint4*>(L.stage + src));
    }
    const u32 pq = P | (sz << 16), oq = off | (hn << 16);
    for (u32 g = 0; g < k; g++) {
        const u32 a = (u32)__builtin_amdgcn_readlane((int)pq, (int)g), b = (u32)__builtin_amdgcn_readlane((int)oq, (int)g);
        const u32 Pg = a & 0xffffu, szg = a >> 16, og = b & 0xffffu, hg = b >> 16;
        const u32 tail0 = hg + (((szg - hg) >> 4) << 4), tg = (szg - hg) & 15u;
        if (lane < hg) gseds[og + lane] = L.stage[Pg + lane];
        else if (lane >= 16u && lane - 16u < tg) gseds[og + tail0 + lane - 16u] = L.stage[Pg + tail0 + lane - 16u];
    }
    asm volatile("" ::: "memory");
    return total;
}

// ---- the lean emitter for segments of up to four strings (2-bit group ids): the common case ----------------
// Rows are split three ways: rows 0..127 and the TAIL rows tb..S-1 (tb = min(992, S rounded down to 16): at most 32
// rows, ids that may have five bytes) are placed one row per lane with packed DPP scans; rows 128..tb-1 are whole
// lanes of sixteen 4-byte tokens: no per-row conditions, 2 VALU + ds_add_rtn + ds_write_b32 per id.
struct alignas(1024) EmitLds2 {
    u32 tab[4 * 64];                       // [string][lane]: LDS address of this lane's next id of that string
    u32 gt[96];                            // per string: [0..3] LDS address of its ids, [16..19] of its ids >= 129, [32..35] of
                                           // its tail ids; [48..51] / [52..55] source / destination of its full 16-byte
                                           // chunks; [64..87] the ragged ends: (source, destination, bytes) x 8
    alignas(16) uint8_t stage[EM_STAGE + EM_TRASH];
};
struct EmitConst2 {                        // lane constants of the lean emitter
    u32 tokc[16];                          // "ddd," of rows 16*lane .. +15
    u32 htok0, htok1, ttok;                // tokens of rows lane, 64 + lane, tb + lane (first four bytes)
    u32 tsrc, tsh, tlt;                    // tail row: owning lane, shift of its 2-bit field, token length (0: no such row)
    u32 binc, bmask, trash;                // whole-lane rows: 4 / 0, ~0 / 0 (is this lane one of them), LDS address of its dummy dword
    u32 nbody;                             // number of those lanes (uniform)
};
__device__ __forceinline__ u32 lds_add_rtn(u32 addr, u32 inc)
{
    u32 r;
    asm volatile("ds_add_rtn_u32 %0, %1, %2" : "=v"(r) : "v"(addr), "v"(inc) : "memory");
    return r;
}
__device__ __forceinline__ void lds_write32(u32 addr, u32 v) { asm volatile("ds_write_b32 %0, %1" :: "v"(addr), "v"(v) : "memory"); }

template <class PreFlush>
__device__ __forceinline__ void emit_ids2(u32 x0, u32 k, u32 S, u32 lane, const EmitConst2& C, EmitLds2& L, uint8_t* gseds,
                                          PreFlush pre_flush)
{
    const u32 sbase = (u32)(uintptr_t)L.stage, tbase = (u32)(uintptr_t)L.tab;   // LDS byte addresses
    // ---- the rows placed one per lane: rows lane, 64 + lane and the tail row tb + lane
    const u32 f = lane & 15u, src = lane >> 4;
    const u32 g0 = (lane_read(x0, src) >> (2u * f)) & 3u, g1 = (lane_read(x0, src + 4u) >> (2u * f)) & 3u;
    const u32 gt_ = (lane_read(x0, C.tsrc) >> C.tsh) & 3u;
    const bool v0 = lane < S, v1 = lane + 64u < S, vt = C.tlt != 0;
    const u32 tl0 = lane < 9u ? 2u : 3u, tl1 = lane < 35u ? 3u : 4u;       // ids 1-9 | 10-64 and 65-99 | 100-128
    const u32 f0 = v0 ? tl0 << (g0 * 8u) : 0u, f1 = v1 ? tl1 << (g1 * 8u) : 0u, ft = C.tlt << (gt_ * 8u);
    // ---- the whole lanes: ids of strings 0..2 in this lane (string 3 has the rest), prefix over the lanes
    const u32 lo = x0 & 0x55555555u, hi = (x0 >> 1) & 0x55555555u;
    const u32 c1 = (u32)__builtin_popcount(lo & ~hi), c2 = (u32)__builtin_popcount(hi & ~lo), c3 = (u32)__builtin_popcount(lo & hi);
    const u32 pk = ((16u - c1 - c2 - c3) | (c1 << 10) | (c2 << 20)) & C.bmask;
    u32 i0 = f0, i1 = f1, it = ft, sp = pk;
    wave_scan_incl4(i0, i1, it, sp);
    const u32 ex0 = ((i0 - f0) >> (g0 * 8u)) & 0xffu, ex1 = ((i1 - f1) >> (g1 * 8u)) & 0xffu, ext = ((it - ft) >> (gt_ * 8u)) & 0xffu;
    const u32 tot0 = (u32)__builtin_amdgcn_readlane((int)i0, 63), tot1 = (u32)__builtin_amdgcn_readlane((int)i1, 63);
    const u32 tott = (u32)__builtin_amdgcn_readlane((int)it, 63);
    const u32 ep = sp - pk;
    const u32 totp = (u32)__builtin_amdgcn_readlane((int)sp, 63);
    const u32 pre0 = ep & 0x3ffu, pre1 = (ep >> 10) & 0x3ffu, pre2 = ep >> 20, pre3 = 16u * (lane - 8u) - pre0 - pre1 - pre2;
    const u32 t0 = totp & 0x3ffu, t1 = (totp >> 10) & 0x3ffu, t2 = totp >> 20, t3 = 16u * C.nbody - t0 - t1 - t2;
    // ---- geometry of string `lane` (lanes < k): bytes of its ids up to 128 / of its whole-lane ids / of its tail ids
    const u32 sh8 = (lane & 3u) * 8u;
    const u32 hb = ((tot0 >> sh8) & 0xffu) + ((tot1 >> sh8) & 0xffu), tb_ = (tott >> sh8) & 0xffu;
    const u32 bb = 4u * (lane == 0u ? t0 : lane == 1u ? t1 : lane == 2u ? t2 : t3);
    // The string's region in `stage` starts at a multiple of 16 and has `lead` bytes of padding in front, so that
    // (P + 1 + hb) % 4 == 0: the 4-byte tokens are dword-aligned.  It is copied out as hn bytes up to the first 16-byte
    // boundary, nfull aligned 16-byte chunks and tn bytes behind them.  One scan gives region, output offset and chunk index.
    const u32 sz = lane < k ? 1u + hb + bb + tb_ : 0u;           // '{' + tokens; the last ',' becomes '}'
    const u32 lead = (3u - hb) & 3u;
    const u32 hn = lead ? (sz < 16u - lead ? sz : 16u - lead) : 0u;
    const u32 nfull = (sz - hn) >> 4, tn = (sz - hn) & 15u;
    const u32 rs = lane < k ? (lead + sz + 15u) >> 4 : 0u;       // region in units of 16 bytes
    const u32 pq = rs | (sz << 9) | (nfull << 22);               // sums: <= 290 | <= 4125 | <= 260
    const u32 sq = wave_scan_incl(pq), eq_ = sq - pq;
    const u32 P = ((eq_ & 0x1ffu) << 4) + lead;
    const u32 off = (eq_ >> 9) & 0x1fffu, cs = eq_ >> 22;
    const u32 T = (u32)__builtin_amdgcn_readlane((int)sq, 63) >> 22;
    if (lane < k) { L.gt[lane] = sbase + P + 1u; L.gt[16 + lane] = sbase + P + 1u + hb; L.gt[32 + lane] = sbase + P + 1u + hb + bb; }
    // ---- cursors of the whole lanes (the other lanes: their dummy dword, advanced by 0)
    {
        const uint4 b4 = *reinterpret_cast<const uint4*>(&L.gt[16]);
        const bool body = C.bmask != 0;
        L.tab[lane] = body ? b4.x + 4u * pre0 : C.trash;
        L.tab[64 + lane] = body ? b4.y + 4u * pre1 : C.trash;
        L.tab[128 + lane] = body ? b4.z + 4u * pre2 : C.trash;
        L.tab[192 + lane] = body ? b4.w + 4u * pre3 : C.trash;
    }
    // ---- tokens of rows 0..127 and of the tail rows (single bytes: any alignment)
    if (v0) {
        const u32 a = L.gt[g0] + ex0;
        lds_put2(a, C.htok0);
        if (lane >= 9u) lds_put1<2>(a, C.htok0 >> 16);
    }
    if (v1) {
        const u32 a = L.gt[g1] + ((tot0 >> (g1 * 8u)) & 0xffu) + ex1;
        lds_put2(a, C.htok1);
        lds_put1<2>(a, C.htok1 >> 16);
        if (lane >= 35u) lds_put1<3>(a, C.htok1 >> 24);
    }
    if (vt) {
        const u32 a = L.gt[32 + gt_] + ext;
        lds_put2(a, C.ttok);
        lds_put1<2>(a, C.ttok >> 16);
        lds_put1<3>(a, C.ttok >> 24);
        if (C.tlt == 5u) lds_put1<4>(a, (u32)',');
    }
    // ---- tokens of the whole lanes: 16 returning ds_add in flight, then 16 aligned ds_write_b32
    {
        const u32 lb = tbase + lane * 4u;                     // tab is 1024-aligned: (g << 8) | lb addresses tab[g][lane]
        u32 a[16];
#pragma unroll
        for (int j = 0; j < 16; j++) a[j] = lds_add_rtn((((x0 >> (2 * j)) & 3u) << 8) | lb, C.binc);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                                              "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15])
                     :: "memory");
#pragma unroll
        for (int j = 0; j < 16; j++) lds_write32(a[j], C.tokc[j]);
    }
    // ---- braces (after the tokens: the closing one replaces the last ',')
    if (lane < k) { L.stage[P] = '{'; L.stage[P + sz - 1u] = '}'; }
    asm volatile("" ::: "memory");
    // ---- copy the regions out.  String g: LDS [P, P + sz) -> gseds + off.  hn bytes up to the first 16-byte
    // boundary of the LDS image, nfull aligned 16-byte chunks, tn bytes behind them.
    pre_flush();                                                 // (the caller's wait for its prefetched loads)
    const u32 cs1 = (u32)__builtin_amdgcn_readlane((int)cs, 1), cs2 = (u32)__builtin_amdgcn_readlane((int)cs, 2);
    const u32 cs3 = (u32)__builtin_amdgcn_readlane((int)cs, 3);           // (lanes >= k: no chunks, cs = T)
    if (lane < k) {
        L.gt[48 + lane] = P + hn - 16u * cs; L.gt[52 + lane] = off + hn - 16u * cs;
        u32* e = &L.gt[64 + 6u * lane];
        e[0] = P; e[1] = off; e[2] = hn;
        e[3] = P + hn + 16u * nfull; e[4] = off + hn + 16u * nfull; e[5] = tn;
    }
    for (u32 t = lane; t < T; t += 64u) {
        const u32 g = (t >= cs1 ? 1u : 0u) + (t >= cs2 ? 1u : 0u) + (t >= cs3 ? 1u : 0u);
        const u32 s_ = L.gt[48 + g] + 16u * t, d_ = L.gt[52 + g] + 16u * t;
        store16u(gseds + d_, *reinterpret_cast<const uint4*>(L.stage + s_));
    }
#pragma unroll
    for (int r = 0; r < 2; r++) {                                // the ragged ends: region = 16 lanes, lane = byte
        if ((u32)r * 2u < k) {
            const u32 reg = (u32)r * 4u + (lane >> 4);
            if (reg < 2u * k) {
                const u32* e = &L.gt[64 + 3u * reg];
                const u32 i = lane & 15u;
                if (i < e[2]) gseds[e[1] + i] = L.stage[e[0] + i];
            }
        }
    }
    asm volatile("" ::: "memory");
}

// what the emitter needs of a segment's record, requested one segment ahead
struct EmitRec { uint4 x; u32 hv, rep, tb; u64 cm; };

// the segments of up to four strings (records of the column scan: text in the record; records of k_seg_group:
// text from the first rows in vc)
__global__ void __launch_bounds__(256, 5) k_emit_fast2(FastParams p)
{
    __shared__ EmitLds2 lds_all[4];
    const MsaView& mv = p.mv;
    if (mv.hdr->status) return;
    const u32 lane = threadIdx.x & 63, wv = uniform32(threadIdx.x >> 6);
    EmitLds2& L = lds_all[wv];
    const u32 S = mv.S;
    const u32 tb = S <= 128u ? S : ((S >> 4) << 4 < 992u ? ((S >> 4) << 4 < 128u ? 128u : (S >> 4) << 4) : 992u);   // first tail row
    EmitConst2 C;
    auto tok4 = [](u32 id) -> u32 {                        // first four bytes of "<id>,"
        if (id >= 1000u) return ('0' + id / 1000u) | (('0' + (id / 100u) % 10u) << 8) | (('0' + (id / 10u) % 10u) << 16) | (('0' + id % 10u) << 24);
        if (id >= 100u) return ('0' + id / 100u) | (('0' + (id / 10u) % 10u) << 8) | (('0' + id % 10u) << 16) | ((u32)',' << 24);
        if (id >= 10u) return ('0' + id / 10u) | (('0' + id % 10u) << 8) | ((u32)',' << 16);
        return ('0' + id) | ((u32)',' << 8);
    };
#pragma unroll
    for (int j = 0; j < 16; j++) C.tokc[j] = tok4(lane * 16u + j + 1u);
    C.htok0 = tok4(lane + 1u); C.htok1 = tok4(lane + 65u);
    {
        const u32 row = tb + lane;
        const bool has = row < S;
        C.tsrc = has ? row >> 4 : 0u; C.tsh = 2u * (row & 15u);
        C.tlt = has ? (row + 1u >= 1000u ? 5u : 4u) : 0u;
        C.ttok = tok4(row + 1u);
    }
    C.nbody = tb >= 128u ? (tb >> 4) - 8u : 0u;
    C.bmask = lane >= 8u && lane < (tb >> 4) ? ~0u : 0u;
    C.binc = C.bmask ? 4u : 0u;
    C.trash = (u32)(uintptr_t)L.stage + (u32)EM_STAGE + 4u * lane;
    const u32 nl = (S + 15u) >> 4;
    const u64 nseg = *p.nseg_ptr;
    const u64 p0 = mv.vbit(0) ? 0 : 1;                      // variant and common segments alternate
    const u64 nvs = nseg > p0 ? (nseg - p0 + 1) / 2 : 0;
    const u64 nw = ((u64)gridDim.x * blockDim.x) >> 6;
    auto mine = [](u64 meta) -> bool { return (meta & META_REC) && !(meta & (META_KIND4 | META_KIND8)); };
    auto load_rec = [&](u64 meta) -> EmitRec {
        EmitRec r;
        r.x = make_uint4(0, 0, 0, 0); r.hv = 0; r.rep = 0; r.tb = 0; r.cm = 0;
        if (mine(meta)) {
            if (meta & META_INLINE) {                        // grouped by the column scan
                const uint8_t* rec = p.recf + (meta & META_RECID) * (u64)p.recf_stride;
                if (lane < nl) r.x.x = *reinterpret_cast<const u32*>(rec + lane * 4u);
                r.hv = *reinterpret_cast<const u32*>(rec + p.recf_gid);
                r.tb = rec[p.recf_gid + 4u + lane];
            } else {
                const uint8_t* rec = p.rec + (meta & META_RECID) * (u64)p.rec_stride;
                if (lane < nl) r.x.x = *reinterpret_cast<const u32*>(rec + lane * 4u);
                const uint8_t* hdr = rec + p.rec_gid;
                r.hv = *reinterpret_cast<const u32*>(hdr);
                r.cm = *reinterpret_cast<const u64*>(hdr + REC_H_SLOT);
                r.rep = *reinterpret_cast<const uint16_t*>(hdr + REC_H_REP + lane * 2u);
            }
        }
        return r;
    };
    // Software pipeline over the wave's segments vi, vi+nw, ...: vmcnt retires in issue order, so a wait for a
    // prefetched record also waits for every store issued before it.  The record of the next segment (and the
    // descriptor of the one after it) is therefore requested first, the id text of this segment is built in LDS
    // (no global traffic), and only then the wave waits for the prefetch and issues this segment's stores.
    u64 vi = (u64)blockIdx.x * (blockDim.x >> 6) + wv;
    const u64 v0i = vi < nvs ? vi : 0, v1i = vi + nw < nvs ? vi + nw : v0i;
    u64 meta = nvs ? uniform64(p.segmeta[2 * v0i + p0]) : 0, meta_n = nvs ? uniform64(p.segmeta[2 * v1i + p0]) : 0;
    u64 qoff = nvs ? uniform64(p.seds_len[2 * v0i + p0]) : 0, qoff_n = nvs ? uniform64(p.seds_len[2 * v1i + p0]) : 0;
    u64 eoff = nvs ? uniform64(p.eds_len[2 * v0i + p0]) : 0, eoff_n = nvs ? uniform64(p.eds_len[2 * v1i + p0]) : 0;
    EmitRec rc = load_rec(meta);
    while (vi < nvs) {
        const u64 seg = 2 * vi + p0;
        const u64 v2 = vi + 2 * nw < nvs ? vi + 2 * nw : vi;
        const EmitRec rc_n = load_rec(vi + nw < nvs ? meta_n : 0);
        const u64 meta_v = p.segmeta[2 * v2 + p0];          // same address in every lane; made scalar
        const u64 qoff_v = p.seds_len[2 * v2 + p0];         // only after the wait below
        const u64 eoff_v = p.eds_len[2 * v2 + p0];
        auto pre_flush = [&]() {
            asm volatile("" :: "v"(rc_n.x.x), "v"(rc_n.hv), "v"(rc_n.rep), "v"(rc_n.tb), "v"(rc_n.cm), "v"(meta_v), "v"(qoff_v), "v"(eoff_v));
        };
        if (mine(meta)) {
            const u32 hdr0 = uniform32(rc.hv);
            const u32 k = hdr0 & 0xffu;
            uint8_t* e = p.eds + eoff;
            if (meta & META_INLINE) {
                if (lane < ((hdr0 >> 8) & 0xffu)) e[lane] = (uint8_t)rc.tb;
            } else {                                         // "{" s0 "," s1 ... "}" from the first rows of the strings
                const u32 ncol = (hdr0 >> 16) & 0xffu;
                const u64 cm = uniform64(rc.cm);
                const u64 slot0 = cm & CNT_SLOT;
                const bool scatter = (cm & CNT_SCATTER) != 0, mixed = (cm & CNT_MIXED) != 0;
                const u64 seg_a = (scatter || mixed) ? uniform64(p.seg_start[seg]) : 0;
                auto cell = [&](u32 c, u32 r) -> u32 {         // byte of row r in column c of the segment
                    if (mixed && !mv.vbit(seg_a + c)) return mv.ref_byte(seg_a + c);
                    const u64 sl = (scatter || mixed) ? mv.slot(seg_a + c) : slot0 + c;
                    return mv.vc[sl * (u64)mv.Spad + r];
                };
                const u32 rep_l = lane < k ? rc.rep : 0u;
                if (k * ncol <= 64u) {                         // lane = (string, column): one load round trip
                    const u32 g = lane / ncol, c = lane - g * ncol;
                    const u32 r = (u32)__shfl((int)rep_l, (int)(g < k ? g : 0u), 64);
                    u32 ch = 0;
                    if (g < k) {
                        ch = cell(c, r);
                        if (ch == '-' || ch == '\n') ch = 0;
                    }
                    const u64 m = ballot64(ch != 0);
                    if (lane == 0) e[0] = '{';
                    if (ch) e[1 + g + mbcnt(m)] = (uint8_t)ch;
                    if (g < k && c == 0) {                     // separator after string g's letters
                        const u32 endl = (g + 1) * ncol;
                        const u64 upto = endl >= 64u ? ~0ull : ((1ull << endl) - 1);
                        e[1 + g + (u32)__builtin_popcountll(m & upto)] = (g + 1 < k) ? ',' : '}';
                    }
                } else {
                    if (lane == 0) e[0] = '{';
                    u32 eo = 1;
                    for (u32 g = 0; g < k; g++) {              // lane = column: the first row's letters
                        const u32 r = (u32)__builtin_amdgcn_readlane((int)rep_l, (int)g);
                        u32 ch = 0;
                        if (lane < ncol) {
                            ch = cell(lane, r);
                            if (ch == '-' || ch == '\n') ch = 0;
                        }
                        const u64 m = ballot64(ch != 0);
                        const u32 len = (u32)__builtin_popcountll(m);
                        if (ch) e[eo + mbcnt(m)] = (uint8_t)ch;
                        if (lane == 0) e[eo + len] = (g + 1 < k) ? ',' : '}';
                        eo += len + 1;
                    }
                }
            }
            emit_ids2(rc.x.x, k, S, lane, C, L, p.seds + qoff, pre_flush);
        } else pre_flush();
        vi += nw;
        rc = rc_n; meta = meta_n; qoff = qoff_n; eoff = eoff_n;
        meta_n = uniform64(meta_v); qoff_n = uniform64(qoff_v); eoff_n = uniform64(eoff_v);
    }
}

// WIDE false: the segments of up to four strings (2-bit group ids); true: those of 5..64 strings
// KMAX (WIDE): 8 - the segments of 5..8 strings, at four waves per SIMD (128 VGPRs); 16 - those of 9..64 strings (17..64
// sixteen at a time) at two.  A work list each (p.wide_list / p.wide16_list).  (Round 3, each kernel alone on the
// machine: one launch for all of them 1.28 ms; the two instantiations walking ONE list and skipping what is the
// other's 0.55 + 0.90 ms.)
template <bool HAS5, bool WIDE, int KMAX = 16>
__global__ void __launch_bounds__(256, WIDE ? (KMAX == 8 ? 4 : 2) : 5) k_emit_fast(FastParams p)
{
    using WaveLds = EmitWaveLdsT<WIDE ? KMAX + 1 : 5>;
    __shared__ WaveLds lds_all[4];
    const MsaView& mv = p.mv;
    if (mv.hdr->status) return;
    const u32 lane = threadIdx.x & 63, wv = uniform32(threadIdx.x >> 6);
    WaveLds& L = lds_all[wv];
    const u32 S = mv.S;
    L.tab[(WIDE ? KMAX : 4) * 64 + lane] = (u32)EM_STAGE + 4u * lane;   // dummy cursors (never advanced: they are added 0)
    // lane constants: tokens "ddd," of this lane's rows 16*lane .. +15 (ids 100..999; "dddd" from 1000), of rows
    // `lane` and `64 + lane`, and which of this lane's rows are placed by the owning lane (rows >= 128)
    u32 tokc[16];
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const u32 id = lane * 16u + j + 1u;
        u32 t;
        if (id >= 1000u) t = ('0' + id / 1000u) | (('0' + (id / 100u) % 10u) << 8) | (('0' + (id / 10u) % 10u) << 16) | (('0' + id % 10u) << 24);
        else t = ('0' + id / 100u) | (('0' + (id / 10u) % 10u) << 8) | (('0' + id % 10u) << 16) | ((u32)',' << 24);
        tokc[j] = t;
    }
    u32 htok0, htok1;
    {
        const u32 a = lane + 1u, b = lane + 65u;
        htok0 = a < 10u ? ('0' + a) | ((u32)',' << 8) : ('0' + a / 10u) | (('0' + a % 10u) << 8) | ((u32)',' << 16);
        htok1 = b < 100u ? ('0' + b / 10u) | (('0' + b % 10u) << 8) | ((u32)',' << 16)
                         : ('0' + b / 100u) | (('0' + (b / 10u) % 10u) << 8) | (('0' + b % 10u) << 16) | ((u32)',' << 24);
    }
    const u32 nvb = lane >= 8u && S > lane * 16u ? (S - lane * 16u < 16u ? S - lane * 16u : 16u) : 0u;
    const u32 amv = (1u << nvb) - 1u;                       // rows >= 128 of this lane that exist
    const bool hv0 = lane < S, hv1 = lane + 64u < S;
    const u32 nl = (S + 15u) >> 4;
    const u64 nseg = *p.nseg_ptr;
    const u64 p0 = mv.vbit(0) ? 0 : 1;                      // variant and common segments alternate
    const u64 nvs = nseg > p0 ? (nseg - p0 + 1) / 2 : 0;
    const u64 nw = ((u64)gridDim.x * blockDim.x) >> 6;
    auto load_rec = [&](u64 meta) -> EmitRec {
        EmitRec r;
        r.x = make_uint4(0, 0, 0, 0); r.hv = 0; r.rep = 0; r.tb = 0; r.cm = 0;
        if ((meta & META_REC) && (meta & META_INLINE)) {     // grouped by the column scan: text in the record
            if (WIDE == ((meta & META_KIND4) != 0)) {
                const uint8_t* rec = p.recf + (meta & META_RECID) * (u64)p.recf_stride;
                if (lane < nl) {
                    if (WIDE) { const uint2 v = *reinterpret_cast<const uint2*>(rec + lane * 8u); r.x.x = v.x; r.x.y = v.y; }
                    else r.x.x = *reinterpret_cast<const u32*>(rec + lane * 4u);
                }
                r.hv = *reinterpret_cast<const u32*>(rec + p.recf_gid);
                r.tb = rec[p.recf_gid + 4u + lane];
            }
        } else if ((meta & META_REC) && ((meta & (META_KIND4 | META_KIND8)) != 0) == WIDE) {
            const uint8_t* rec = p.rec + (meta & META_RECID) * (u64)p.rec_stride;
            if (lane < nl) {
                if (meta & META_KIND8) r.x = *reinterpret_cast<const uint4*>(rec + lane * 16u);
                else if (meta & META_KIND4) { const uint2 v = *reinterpret_cast<const uint2*>(rec + lane * 8u); r.x.x = v.x; r.x.y = v.y; }
                else r.x.x = *reinterpret_cast<const u32*>(rec + lane * 4u);
            }
            const uint8_t* hdr = rec + p.rec_gid;
            r.hv = *reinterpret_cast<const u32*>(hdr);
            r.cm = *reinterpret_cast<const u64*>(hdr + REC_H_SLOT);
            r.rep = *reinterpret_cast<const uint16_t*>(hdr + REC_H_REP + lane * 2u);
        }
        return r;
    };
    // Software pipeline over the wave's segments vi, vi+nw, ...: vmcnt retires in issue order, so a wait for a
    // prefetched record also waits for every store issued before it.  The record of the next segment (and the
    // descriptor of the one after it) is therefore requested first, the id text of this segment is built in LDS
    // (no global traffic), and only then the wave waits for the prefetch and issues this segment's stores.
    // the wide emitter walks its work list (few segments: no software pipeline)
    const u64* const wlist = KMAX == 8 ? p.wide_list : p.wide16_list;
    const u64 nwide = KMAX == 8 ? *p.wide_count : *p.wide16_count;
    for (u64 it = (u64)blockIdx.x * (blockDim.x >> 6) + wv; it < nwide; it += nw) {
        const u64 vi = uniform64(wlist[it]);
        const u64 seg = 2 * vi + p0;
        const u64 meta = uniform64(p.segmeta[seg]);
        const u64 qoff = uniform64(p.seds_len[seg]), eoff = uniform64(p.eds_len[seg]);
        const EmitRec rc = load_rec(meta);
        const EmitRec rc_n = rc;
        const u64 meta_v = 0, qoff_v = 0, eoff_v = 0;
        const u32 hdr0 = uniform32(rc.hv);
        const u32 k = hdr0 & 0xffu, textlen = (hdr0 >> 8) & 0xffu, ncol = (hdr0 >> 16) & 0xffu;
        const bool fast = (meta & META_REC) != 0 && ((meta & (META_KIND4 | META_KIND8)) != 0) == WIDE &&
                          (!WIDE || (KMAX == 8) == (k <= 8u));
        uint8_t* gseds = p.seds + qoff;
        // the wait for the prefetched record: called right before this segment's id lists are stored (everything
        // before that point that reads global memory is older than the prefetch or was waited for already)
        auto pre_flush = [&]() {
            asm volatile("" :: "v"(rc_n.x.x), "v"(rc_n.x.y), "v"(rc_n.x.z), "v"(rc_n.x.w), "v"(rc_n.hv), "v"(rc_n.rep),
                               "v"(rc_n.tb), "v"(rc_n.cm), "v"(meta_v), "v"(qoff_v), "v"(eoff_v));
        };
        if (fast) {
            // ---- eds: "{" s0 "," s1 ... "}"
            uint8_t* e = p.eds + eoff;
            if (meta & META_INLINE) {
                if (lane < textlen) e[lane] = (uint8_t)rc.tb;
            } else {
                const u64 cm = uniform64(rc.cm);
                const u64 slot0 = cm & CNT_SLOT;
                const bool scatter = (cm & CNT_SCATTER) != 0, mixed = (cm & CNT_MIXED) != 0;
                const u64 seg_a = (scatter || mixed) ? uniform64(p.seg_start[seg]) : 0;
                auto cell = [&](u32 c, u32 r) -> u32 {         // byte of row r in column c of the segment
                    if (mixed && !mv.vbit(seg_a + c)) return mv.ref_byte(seg_a + c);
                    const u64 sl = (scatter || mixed) ? mv.slot(seg_a + c) : slot0 + c;
                    return mv.vc[sl * (u64)mv.Spad + r];
                };
                const u32 rep_l = lane < k ? rc.rep : 0u;
                if (k * ncol <= 64u) {                         // lane = (string, column): one load round trip
                    const u32 g = lane / ncol, c = lane - g * ncol;
                    const u32 r = (u32)__shfl((int)rep_l, (int)(g < k ? g : 0u), 64);
                    u32 ch = 0;
                    if (g < k) {
                        ch = cell(c, r);
                        if (ch == '-' || ch == '\n') ch = 0;
                    }
                    const u64 m = ballot64(ch != 0);
                    if (lane == 0) e[0] = '{';
                    if (ch) e[1 + g + mbcnt(m)] = (uint8_t)ch;
                    if (g < k && c == 0) {                     // separator after string g's letters
                        const u32 endl = (g + 1) * ncol;
                        const u64 upto = endl >= 64u ? ~0ull : ((1ull << endl) - 1);
                        e[1 + g + (u32)__builtin_popcountll(m & upto)] = (g + 1 < k) ? ',' : '}';
                    }
                } else {
                    if (lane == 0) e[0] = '{';
                    u32 eo = 1;
                    for (u32 g = 0; g < k; g++) {              // lane = column: the first row's letters
                        const u32 r = (u32)__builtin_amdgcn_readlane((int)rep_l, (int)g);
                        u32 ch = 0;
                        if (lane < ncol) {
                            ch = cell(lane, r);
                            if (ch == '-' || ch == '\n') ch = 0;
                        }
                        const u64 m = ballot64(ch != 0);
                        const u32 len = (u32)__builtin_popcountll(m);
                        if (ch) e[eo + mbcnt(m)] = (uint8_t)ch;
                        if (lane == 0) e[eo + len] = (g + 1 < k) ? ',' : '}';
                        eo += len + 1;
                    }
                }
            }
            if (WIDE && (meta & META_KIND8)) { if constexpr (KMAX == 16) {
                // 17..64 strings: sixteen at a time (strings 16t .. 16t+15 of the rows whose id is in that range)
                const u32 f = lane & 15u, src = lane >> 4;
                auto head_gid = [&](u32 sl) -> u32 {        // (all four reads by all lanes: ds_bpermute takes data from active lanes only)
                    const u32 dx = lane_read(rc.x.x, sl), dy = lane_read(rc.x.y, sl), dz = lane_read(rc.x.z, sl), dw = lane_read(rc.x.w, sl);
                    const u32 d = (f >> 2) == 0 ? dx : (f >> 2) == 1 ? dy : (f >> 2) == 2 ? dz : dw;
                    return (d >> ((f & 3u) * 8u)) & 0xffu;
                };
                const u32 G0 = head_gid(src), G1 = head_gid(src + 4u);
                const uint4 ones = make_uint4(~0u, ~0u, ~0u, ~0u);
                const uint2 xx = pack_gid4(rc.x, ones);
                u32 run = 0;
                for (u32 t = 0; t * 16u < k; t++) {
                    const u32 tt = t * 0x01010101u;
                    const u32 am = (eq_byte4((rc.x.x >> 4) & 0x0f0f0f0fu, tt) | (eq_byte4((rc.x.y >> 4) & 0x0f0f0f0fu, tt) << 4) |
                                    (eq_byte4((rc.x.z >> 4) & 0x0f0f0f0fu, tt) << 8) | (eq_byte4((rc.x.w >> 4) & 0x0f0f0f0fu, tt) << 12)) & amv;
                    const u32 kt = k - t * 16u < 16u ? k - t * 16u : 16u;
                    run += emit_ids<WIDE ? 4 : 2, HAS5, 16>(xx.x, xx.y, am, G0 & 15u, G1 & 15u, hv0 && (G0 >> 4) == t, hv1 && (G1 >> 4) == t,
                                             kt, S, lane, tokc, htok0, htok1, L, gseds + run, pre_flush);
                }
            } } else if (WIDE) {
                const u32 f = lane & 15u, src = lane >> 4;
                const u32 a0 = lane_read(rc.x.x, src), a1 = lane_read(rc.x.y, src), b0 = lane_read(rc.x.x, src + 4u), b1 = lane_read(rc.x.y, src + 4u);
                const u32 g0 = (((f & 8u) ? a1 : a0) >> (4u * (f & 7u))) & 15u, g1 = (((f & 8u) ? b1 : b0) >> (4u * (f & 7u))) & 15u;
                emit_ids<WIDE ? 4 : 2, HAS5, KMAX>(rc.x.x, rc.x.y, amv, g0, g1, hv0, hv1, k, S, lane, tokc, htok0, htok1, L, gseds, pre_flush);
            } else {
                const u32 f = lane & 15u, src = lane >> 4;
                const u32 g0 = (lane_read(rc.x.x, src) >> (2u * f)) & 3u, g1 = (lane_read(rc.x.x, src + 4u) >> (2u * f)) & 3u;
                emit_ids<WIDE ? 2 : 2, HAS5, 4>(rc.x.x, 0u, amv, g0, g1, hv0, hv1, k, S, lane, tokc, htok0, htok1, L, gseds, pre_flush);
            }
        } else pre_flush();
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static u64 token_total(u32 S)
{
    u64 t = 0;
    for (u32 r = 1; r <= S; r++) { u32 d = 1; for (u32 v = r; v >= 10; v /= 10) d++; t += d + 1; }
    return t;
}

void MsaPipeline::launch_timer_begin(const char* name, hipStream_t st)
{
    if (!timing_) return;
    TimedKernel tk; tk.name = name;
    EDSX_HIP(hipEventCreate(&tk.t0)); EDSX_HIP(hipEventCreate(&tk.t1));
    EDSX_HIP(hipEventRecord(tk.t0, st));
    timed_.push_back(tk);
}
void MsaPipeline::launch_timer_end(hipStream_t st)
{
    if (!timing_) return;
    EDSX_HIP(hipEventRecord(timed_.back().t1, st));
}
// harvest the finished event pairs of the previous call into the per-kernel accumulators
void MsaPipeline::clear_timers()
{
    for (auto& t : timed_) {
        float v = 0;
        (void)hipEventSynchronize(t.t1);
        if (hipEventElapsedTime(&v, t.t0, t.t1) == hipSuccess) {
            bool found = false;
            for (auto& a : acc_) if (a.name == t.name) { a.total_ms += v; a.count++; found = true; break; }
            if (!found) acc_.push_back({t.name, v, 1});
        }
        (void)hipEventDestroy(t.t0); (void)hipEventDestroy(t.t1);
    }
    timed_.clear();
}
void MsaPipeline::set_timing(bool on)
{
    clear_timers();
    acc_.clear();
    timing_ = on;
}
// accumulated device time per kernel since set_timing(true): ms[i] = total, counts[i] = launches
int MsaPipeline::get_timing(const char** names, float* ms, int* counts, int cap)
{
    clear_timers();
    int n = 0;
    for (const auto& a : acc_) {
        if (n >= cap) break;
        names[n] = a.name; ms[n] = (float)a.total_ms; counts[n] = a.count; n++;
    }
    return n;
}

#define TIMED(name, st, ...) do { launch_timer_begin(name, st); __VA_ARGS__; launch_timer_end(st); } while (0)

MsaPipeline::~MsaPipeline()
{
    clear_timers();
    if (side_ready_) {
        for (auto& x : side_) (void)hipStreamDestroy(x);
        for (auto& e : side_ev_) (void)hipEventDestroy(e);
    }
}

void MsaPipeline::ensure_side_streams()
{
    if (side_ready_) return;
    // lowest priority: where they run beside the main emitter (EDSX_SIDE=2) they only take what it leaves free
    int least = 0, greatest = 0;
    EDSX_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    for (auto& x : side_) EDSX_HIP(hipStreamCreateWithPriority(&x, hipStreamNonBlocking, least));
    for (auto& e : side_ev_) EDSX_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    side_ready_ = true;
}

static const char* status_message(u64 st)
{
    if (st & ST_NOT_FASTA) return "Invalid MSA: expected a FASTA header line starting with '>'";
    if (st & ST_FEW_ROWS) return "Invalid MSA: at least two sequences are required";
    if (st & ST_TOO_MANY_ROWS) return "MSA has more sequences than this build supports (9999999)";
    if (st & ST_NEWLINE_IN_DATA) return "Invalid MSA: rows must have equal length and a uniform line width";
    if (st & ST_LAYOUT) return "Invalid MSA: rows must have equal length and a uniform line width";
    return "MSA transform failed";
}

template <int T, int RPT, bool HOLD, bool LANEROWS, int MINW, bool ROWS64 = false, bool BIG = false>
static void launch_k1(const K1Params& p, size_t lds, hipStream_t st)
{
    auto kern = k_scan_extract<T, RPT, HOLD, LANEROWS, MINW, ROWS64, BIG>;
    EDSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)p.ntiles), dim3(T), lds, st, p);
}

void MsaPipeline::plan(const uint8_t* d_msa, size_t n, uint32_t l, hipStream_t st,
                       uint64_t* eds_bytes, uint64_t* seds_bytes)
{
    clear_timers();
    planned_ = false;
    file_ = d_msa; n_ = n; l_ = l;
    if (n == 0) throw FormatError("Invalid MSA: empty input");

    hdr_.ensure(sizeof(MsaHdr));
    MsaHdr* dh = hdr_.as<MsaHdr>();
    // ---- K0: geometry + row index (one host sync: everything below is sized from it).  The row table starts with room
    // for ROW_CAP rows; an alignment with more rows is indexed again with the table sized from its first row's length.
    u64 row_cap = ROW_CAP;
    for (int attempt = 0;; attempt++) {
        rows_.ensure(sizeof(u64) * (row_cap + ROW_PAD));
        idx_tmp_.ensure(2 * sizeof(u64) * row_cap);
        EDSX_HIP(hipMemsetAsync(dh, 0, sizeof(MsaHdr), st));
        TIMED("k_find_hdr_end", st, hipLaunchKernelGGL(k_find_hdr_end, dim3(1), dim3(64), 0, st, d_msa, (u64)n, dh));
        TIMED("k_find_row0", st, hipLaunchKernelGGL(k_find_row0, dim3(512), dim3(1024), 0, st, d_msa, (u64)n, dh));
        TIMED("k_index_spec", st, hipLaunchKernelGGL(k_index_spec, dim3(512), dim3(256), 0, st, d_msa, (u64)n, dh,
                                                     idx_tmp_.as<u64>(), idx_tmp_.as<u64>() + row_cap, row_cap));
        TIMED("k_index_check", st, hipLaunchKernelGGL(k_index_check, dim3(1), dim3(1024), 0, st, d_msa, (u64)n, dh,
                                                      idx_tmp_.as<u64>(), idx_tmp_.as<u64>() + row_cap, rows_.as<u64>(), row_cap));
        TIMED("k_index_rows", st, hipLaunchKernelGGL(k_index_rows, dim3(1), dim3(64), 0, st, d_msa, (u64)n, dh,
                                                     rows_.as<u64>(), row_cap));
        EDSX_HIP(hipMemcpyAsync(&h_, dh, sizeof(MsaHdr), hipMemcpyDeviceToHost, st));
        EDSX_HIP(hipStreamSynchronize(st));
        if (attempt == 0 && h_.status == ST_TOO_MANY_ROWS) {
            const u64 most = n / (h_.Draw + 3) + 2;            // ">\n" + the row + "\n" at least
            if (most > row_cap && row_cap <= MAX_ROWS) { row_cap = std::min<u64>(most, MAX_ROWS + 1); clear_timers(); continue; }
        }
        break;
    }
    if (h_.status & ST_TOO_MANY_ROWS) throw LimitError(status_message(ST_TOO_MANY_ROWS));
    if (h_.status) throw FormatError(status_message(h_.status));
    if (h_.S > MAX_ROWS) throw LimitError(status_message(ST_TOO_MANY_ROWS));
    {   // padding for the column scan's 16-row loads (at most 256 threads x 16 rows past the last one)
        const u64 padded = std::min<u64>(row_cap + ROW_PAD, (h_.S + 15) / 16 * 16 + 4096);
        hipLaunchKernelGGL(k_pad_rows, dim3(4), dim3(256), 0, st, rows_.as<u64>(), h_.S, padded);
    }

    const u64 Draw = h_.Draw;
    const u32 Spad = vc_pitch((u32)h_.S);
    {   // first guess for vc: 12.5 % variant columns; grown to the exact need on overflow
        u64 want = std::max<u64>(Draw / 8, 4096);
        if (want > Draw) want = Draw;
        vc_.ensure((size_t)want * Spad);
    }
    for (int attempt = 0;; attempt++) {
        vc_cap_cols_ = vc_.cap / Spad;
        plan_body(st);
        EDSX_HIP(hipMemcpyAsync(&h_, dh, sizeof(MsaHdr), hipMemcpyDeviceToHost, st));
        EDSX_HIP(hipStreamSynchronize(st));
        EDSX_HIP(hipGetLastError());
        if ((h_.status & ST_VC_OVERFLOW) && !(h_.status & ~(u64)ST_VC_OVERFLOW) && attempt == 0) {
            vc_.ensure((size_t)h_.nv * Spad);            // K1 counted every variant column
            EDSX_HIP(hipMemsetAsync(&dh->nv, 0, sizeof(u64), st));
            EDSX_HIP(hipMemsetAsync(&dh->status, 0, sizeof(u64), st));
            clear_timers();
            continue;
        }
        break;
    }
    if (h_.status) throw FormatError(status_message(h_.status & ~(u64)ST_VC_OVERFLOW));
    if (getenv("EDSX_COUNTS"))
        fprintf(stderr, "[edsx] segments %llu variant %llu | grouping list %llu heavy %llu | wide emit %llu | generic %llu + %llu | fused %d\n",
                (unsigned long long)(l == 0 ? h_.R : h_.nseg), (unsigned long long)h_.nvs, (unsigned long long)h_.cnt_n,
                (unsigned long long)h_.heavy_n, (unsigned long long)(h_.wide_n + h_.wide16_n), (unsigned long long)h_.slow_n,
                (unsigned long long)h_.slow_n2, (int)fuse_);
    if (l == 0) h_.nseg = h_.R;
    *eds_bytes = h_.E;
    *seds_bytes = h_.Q;
    planned_ = true;
}

// everything between the row index and the output sizes; no host synchronisation inside
void MsaPipeline::plan_body(hipStream_t st)
{
    MsaHdr* dh = hdr_.as<MsaHdr>();
    const uint8_t* d_msa = file_;
    const uint32_t l = l_;
    const u64 S = h_.S, L = h_.L, lw = h_.lw, Draw = h_.Draw;
    const u32 Spad = vc_pitch((u32)S);
    const u64 nwords = h_.nwords, nwords_raw = h_.nwords_raw;

    // ---- workspace (sized by the worst case "every column starts a run")
    vraw_.ensure(8 * (nwords_raw + 1));
    wslot_.ensure(8 * (nwords_raw + 1));
    if (lw) v_.ensure(8 * (nwords + 1));
    hrun_.ensure(8 * (nwords + 1));
    hseg_.ensure(8 * (nwords + 1));
    cnt_.ensure(8 * (nwords + 1));
    wbase_.ensure(8 * (nwords + 1));
    segbase_.ensure(8 * (nwords + 1));
    scan_tmp_.ensure(8 * 4 * ((L + 2) / SCAN_TILE + 4));      // up to four arrays per pass (exclusive_scan_multi)
    run_start_.ensure(8 * (L + 2));
    if (l) { flag_.ensure(8 * (L + 2)); seg_start_.ensure(8 * (L + 2)); }
    eds_len_.ensure(8 * (L + 2));
    seds_len_.ensure(8 * (L + 2));

    // ---- K1 geometry.  A workgroup of T threads keeps RPT 16-byte chunks per thread in registers:
    // rows per pass = T / CPR, tile width W = 16 * CPR columns.  Two workgroups per CU let one
    // tile's extraction phase overlap the other's load phase.
    // T = 512, RPT = 16: two workgroups per CU (1024 threads x 16 and 1024 x 8 were measured in round 1: slower).
    const int T = 512, RPT = 16;
    u32 cpr_log2 = 8;
    while (cpr_log2 > 2 && (u64)RPT * (T >> cpr_log2) < S) cpr_log2--;
    const bool hold = (u64)RPT * (T >> cpr_log2) >= S;
    const u64 W = 16ull << cpr_log2;
    const u64 ntiles = (Draw + W - 1) / W;
    // 40 KB of column image (39 columns of 1000 rows; denser tiles take the batched path) leave room for a third
    // workgroup per CU to start while the two resident ones finish their grouping tails (-1.7 % on the scan)
    big_ = S > LDS_ROWS;                                  // row tables / column image in HBM instead of LDS
    const size_t colbuf_bytes = big_ ? 64 : (size_t)S * 8 <= 40 * 1024 ? 40 * 1024 : 64 * 1024;
    if (ntiles > 0x7fffffffull) throw FormatError("MSA too large for one launch");

    K1Params kp;
    kp.file = d_msa; kp.row_start = rows_.as<u64>(); kp.hdr = dh;
    kp.Vraw = vraw_.as<u64>(); kp.word_slot = wslot_.as<u64>(); kp.vc = vc_.as<uint8_t>();
    kp.vc_cap_cols = vc_cap_cols_; kp.Draw = Draw; kp.lw = lw; kp.S = (u32)S; kp.Spad = Spad;
    kp.cpr_log2 = cpr_log2; kp.cap_cols = (u32)(colbuf_bytes / Spad); kp.ntiles = ntiles;
    const bool lane_rows = hold && RPT == 16;             // thread rows = 16 consecutive rows = 16 consecutive vc bytes
    // fused grouping: context length 0 (segments = runs), one-line rows (raw position = column), wave-per-segment code
    fuse_ = l == 0 && lw == 0 && S <= 1024 && lane_rows;
    kp.fuse = fuse_ ? 1u : 0u; kp.Fraw = nullptr; kp.rec_info = nullptr; kp.recf = nullptr; kp.recf_stride = 0; kp.recf_gid = 0;
    if (fuse_) {
        recf_gid_ = (8u * (((u32)S + 15u) / 16u) + 15u) & ~15u;
        recf_stride_ = (recf_gid_ + 4u + REC_TEXT_MAX + 63u) & ~63u;
        fraw_.ensure(8 * (nwords_raw + 1));
        rec_info_.ensure(4 * ((size_t)vc_cap_cols_ + 2));
        recf_.ensure(((size_t)vc_cap_cols_ + 2) * recf_stride_);
        kp.Fraw = fraw_.as<u64>(); kp.rec_info = rec_info_.as<u32>(); kp.recf = recf_.as<uint8_t>();
        kp.recf_stride = recf_stride_; kp.recf_gid = recf_gid_;
    }
    launch_timer_begin("k_scan_extract", st);
    if (lane_rows && fuse_ && S <= 64) launch_k1<512, 16, true, true, 4, true>(kp, colbuf_bytes, st);   // one row per lane in the fused grouping
    else if (lane_rows) launch_k1<512, 16, true, true, 4>(kp, colbuf_bytes, st);
    else if (hold) launch_k1<512, 16, true, false, 4>(kp, colbuf_bytes, st);
    else if (big_) launch_k1<512, 16, false, false, 4, false, true>(kp, colbuf_bytes, st);
    else launch_k1<512, 16, false, false, 4>(kp, colbuf_bytes, st);
    launch_timer_end(st);

    const u64* V = lw ? v_.as<u64>() : vraw_.as<u64>();
    if (lw) TIMED("k_vmap", st, hipLaunchKernelGGL(k_vmap, dim3(1024), dim3(256), 0, st, vraw_.as<u64>(),
                                                   v_.as<u64>(), L, lw, nwords));

    // ---- K2: runs -> segments
    u64* d_nwords = &dh->nwords;
    TIMED("k_runstart_words", st, hipLaunchKernelGGL(k_runstart_words, dim3(1024), dim3(256), 0, st, V,
                                                     hrun_.as<u64>(), cnt_.as<u64>(), L, nwords));
    TIMED("scan_runs", st, exclusive_scan_u64(cnt_.as<u64>(), wbase_.as<u64>(), d_nwords, &dh->R,
                                              scan_tmp_.as<u64>(), st));
    TIMED("k_write_runs", st, hipLaunchKernelGGL(k_write_positions, dim3(1024), dim3(256), 0, st, hrun_.as<u64>(),
                                                 wbase_.as<u64>(), run_start_.as<u64>(), nwords, &dh->R, L));
    const u64* seg_start; const u64* Hseg; const u64* segbase; const u64* d_nseg;
    if (l == 0) {
        seg_start = run_start_.as<u64>(); Hseg = hrun_.as<u64>(); segbase = wbase_.as<u64>(); d_nseg = &dh->R;
    } else {
        TIMED("k_seg_flags", st, hipLaunchKernelGGL(k_seg_flags, dim3(1024), dim3(256), 0, st, run_start_.as<u64>(),
                                                    V, &dh->R, (u64)l, flag_.as<u64>()));
        // eds_len_ doubles as scratch for the run-sized scan (it is overwritten by k_seg_count later)
        TIMED("scan_flags", st, exclusive_scan_u64(flag_.as<u64>(), eds_len_.as<u64>(), &dh->R, &dh->nseg,
                                                   scan_tmp_.as<u64>(), st));
        EDSX_HIP(hipMemsetAsync(hseg_.ptr, 0, 8 * (nwords + 1), st));
        TIMED("k_write_segs", st, hipLaunchKernelGGL(k_write_segs, dim3(1024), dim3(256), 0, st, run_start_.as<u64>(),
                                                     flag_.as<u64>(), eds_len_.as<u64>(), &dh->R, &dh->nseg,
                                                     seg_start_.as<u64>(), hseg_.as<u64>(), L));
        TIMED("k_popc_words", st, hipLaunchKernelGGL(k_popc_words, dim3(1024), dim3(256), 0, st, hseg_.as<u64>(),
                                                     cnt_.as<u64>(), nwords));
        TIMED("scan_segwords", st, exclusive_scan_u64(cnt_.as<u64>(), segbase_.as<u64>(), d_nwords, &dh->tmp_total,
                                                      scan_tmp_.as<u64>(), st));
        seg_start = seg_start_.as<u64>(); Hseg = hseg_.as<u64>(); segbase = segbase_.as<u64>(); d_nseg = &dh->nseg;
    }

    // ---- K3 + K4: per-segment sizes, offsets
    mv_.file = d_msa; mv_.row_start = rows_.as<u64>(); mv_.V = V; mv_.Vraw = vraw_.as<u64>();
    mv_.word_slot = wslot_.as<u64>(); mv_.vc = vc_.as<uint8_t>(); mv_.hdr = dh; mv_.L = L; mv_.lw = lw;
    mv_.S = (u32)S; mv_.Spad = Spad; mv_.tileW = (u32)W;
    seg_start_p_ = seg_start; hseg_p_ = Hseg; segbase_p_ = segbase; nseg_p_ = d_nseg;
    seg_lds_ = (size_t)18 * S + 64 + (S <= HT_MAX_ROWS ? HtLds::bytes(ht_size_of((u32)S)) + 16 : 0);
    seg_lds_ = (seg_lds_ + 15) & ~(size_t)15;
    stage_off_ = 0;
    stage_cols_ = 0;
    seg_scratch_stride_ = 0;
    if (big_) {
        // the generic kernels' row tables (20 B per row) and the emitter's walk tables: a slice of HBM per workgroup;
        // as many workgroups as 8 GiB of it allow (32 .. 512)
        seg_lds_ = 64;
        seg_scratch_stride_ = (((SegLdsT<true>::bytes((u32)S) + 15) & ~(size_t)15) + walk_table_bytes<true>((u32)S) + 255) & ~(size_t)255;
        big_grid_ = (unsigned)std::max<u64>(32, std::min<u64>(512, (8ull << 30) / seg_scratch_stride_));
        seg_scratch_.ensure((size_t)big_grid_ * seg_scratch_stride_);
    } else {                                                     // as many columns of a segment as fit beside that (<= STAGE_COLS)
        const size_t budget = (size_t)150 * 1024;
        const size_t maps = 2 * (size_t)STAGE_WMAX + 16;     // column map + reference bytes of the common columns
        constexpr size_t GEN_MINCOLS = 24;
        // two (or three) workgroups per CU if 24 columns and more still fit - 8 for more than 1024 rows, which have no other
        // path (wider segments keep their variant columns + a column map)
        const size_t nblk = (S + 63) / 64, S2 = (S + 1) & ~(size_t)1;
        const size_t walk_cols = (nblk * 384 + S2 * 3 + nblk + 16 + Spad + 7) / ((size_t)Spad + 8);     // the emitter's tables for the .seds walk live there too
        const size_t mincols = S > 1024 ? std::max<size_t>(8, walk_cols) : GEN_MINCOLS;
        size_t use = budget;
        {
            const size_t part = (size_t)(150 / 2 - 1) * 1024;
            if (seg_lds_ + maps + mincols * ((size_t)Spad + 8) <= part) use = part;
        }
        if (seg_lds_ + maps + 4 * ((size_t)Spad + 8) <= use) {
            stage_cols_ = (u32)std::min<size_t>(STAGE_COLS, (use - seg_lds_ - maps) / ((size_t)Spad + 8));
            stage_off_ = (u32)seg_lds_;
            seg_lds_ += stage_cols_offset(stage_cols_) + (size_t)stage_cols_ * Spad;
        }
    }

    // grouping cache of the generic kernels (count -> emit): two regions (one per work list; without lists: both)
    gc_stride_ = gcache_stride_of((u32)S, big_);
    // (with the wave-per-segment kernels in front - up to 1024 rows - only the few segments on their slow lists come here:
    // 256 MiB per list; items beyond the cache are simply grouped again by the emitter)
    gc_region_ = (size_t)std::min<u64>(std::max<u64>((u64)(S <= 1024 ? 256u : 1024u) << 20, 4 * gc_stride_), (L / 2 + 4) * (u64)gc_stride_);
    gc_region_ = gc_region_ / gc_stride_ * gc_stride_;
    gcache_.ensure(2 * gc_region_ + 16);

    const u64 tok_total = token_total((u32)S);
    fast_ = S <= 1024;
    SegParams sp;
    sp.mv = mv_; sp.seg_start = seg_start; sp.nseg_ptr = d_nseg; sp.eds_len = eds_len_.as<u64>();
    sp.seds_len = seds_len_.as<u64>(); sp.tok_total = tok_total; sp.list = nullptr; sp.list_n = nullptr;
    sp.stage_cols = stage_cols_; sp.stage_off = stage_off_;
    long_list_.ensure(8 * (L / LONG_COMMON + 4));             // the long common segments
    sp.long_list = long_list_.as<u64>(); sp.long_count = &dh->long_n;
    EDSX_HIP(hipMemsetAsync(&dh->long_n, 0, sizeof(u64), st));
    sp.scratch = seg_scratch_.as<uint8_t>(); sp.scratch_stride = seg_scratch_stride_;
    EDSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_seg_count<false>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)seg_lds_));
    if (fast_) {
        segmeta_.ensure(8 * (L + 2));
        // list 1: too wide or mixed segments (k_seg_meta), list 2: those the grouping kernel gives up on; together
        // at most all variant segments (<= L/2 + 1)
        slow_list_.ensure(8 * 4 * (L / 2 + 4));            // + the wide emitters' two lists
        cnt_list_.ensure(8 * 11 * (L / 2 + 4));               // work lists of the grouping kernels (ordinal + column descriptor, three times), descriptor and flags per variant segment
        fp_.mv = mv_; fp_.seg_start = seg_start; fp_.nseg_ptr = d_nseg; fp_.segmeta = segmeta_.as<u64>();
        fp_.eds_len = eds_len_.as<u64>(); fp_.seds_len = seds_len_.as<u64>();
        fp_.slow_list = slow_list_.as<u64>(); fp_.slow_count = &dh->slow_n;
        fp_.slow_list2 = slow_list_.as<u64>() + (L / 2 + 4); fp_.slow_count2 = &dh->slow_n2;
        fp_.cnt_vi = cnt_list_.as<u64>(); fp_.cnt_cm = fp_.cnt_vi + (L / 2 + 4); fp_.cnt_n = &dh->cnt_n;
        fp_.heavy_vi = fp_.cnt_cm + (L / 2 + 4); fp_.heavy_cm = fp_.heavy_vi + (L / 2 + 4); fp_.heavy_n = &dh->heavy_n;
        fp_.cnt_meta = fp_.heavy_cm + (L / 2 + 4); fp_.cnt_flag = fp_.cnt_meta + (L / 2 + 4); fp_.wide_flag = fp_.cnt_flag + (L / 2 + 4);
        fp_.heavy_flag = fp_.wide_flag + (L / 2 + 4); fp_.heavy2_vi = fp_.heavy_flag + (L / 2 + 4); fp_.heavy2_cm = fp_.heavy2_vi + (L / 2 + 4);
        fp_.heavy2_n = &dh->heavy2_n;
        fp_.wide16_flag = fp_.heavy2_cm + (L / 2 + 4);
        fp_.wide_list = slow_list_.as<u64>() + 2 * (L / 2 + 4); fp_.wide_count = &dh->wide_n;
        fp_.wide16_list = slow_list_.as<u64>() + 3 * (L / 2 + 4); fp_.wide16_count = &dh->wide16_n;
        fp_.eds = nullptr; fp_.seds = nullptr; fp_.tok_total = tok_total;
        // one record per variant segment; there are at most as many as variant columns
        fp_.rec_stride = rec_stride((u32)S); fp_.rec_gid = rec_gid_bytes((u32)S);
        rec_.ensure(((size_t)std::min<u64>(vc_cap_cols_, L / 2 + 2) + 2) * fp_.rec_stride);
        fp_.rec = rec_.as<uint8_t>();
        fp_.Fraw = fuse_ ? fraw_.as<u64>() : nullptr; fp_.rec_info = rec_info_.as<u32>();
        fp_.recf = recf_.as<uint8_t>(); fp_.recf_stride = recf_stride_; fp_.recf_gid = recf_gid_;
        fp_.long_list = sp.long_list; fp_.long_count = sp.long_count;
        EDSX_HIP(hipMemsetAsync(&dh->slow_n, 0, 2 * sizeof(u64), st));
        EDSX_HIP(hipMemsetAsync(&dh->wide_n, 0, 3 * sizeof(u64), st));     // (wide_n is then set by scan_wide, and added to by k_seg_group)
        EDSX_HIP(hipMemsetAsync(&dh->heavy2_n, 0, 2 * sizeof(u64), st));    // + wide16_n
        TIMED("k_seg_meta", st, hipLaunchKernelGGL(k_seg_meta, dim3(4096), dim3(256), 0, st, fp_));
        {   // list positions of the light / heavy grouping lists and of the wide emitter's list: one pass
            ScanSet<4> ss{{fp_.cnt_flag, fp_.heavy_flag, fp_.wide_flag, fp_.wide16_flag}, {fp_.cnt_flag, fp_.heavy_flag, fp_.wide_flag, fp_.wide16_flag},
                          {&dh->cnt_n, &dh->heavy_n, &dh->wide_n, &dh->wide16_n}};
            TIMED("scan_lists", st, exclusive_scan_multi<4>(ss, &dh->nvs, scan_tmp_.as<u64>(), st));
        }
        TIMED("k_work_scatter", st, hipLaunchKernelGGL(k_work_scatter, dim3(2048), dim3(256), 0, st, fp_, fp_.cnt_flag, fp_.heavy_flag, &dh->nvs));
        // (The light and the heavy grouping kernel and the generic count are independent, but side by side on three
        // streams they only share the machine - measured in round 3: 1.85 ms for the three against 1.33 + 0.41 + 0.01 one
        // after the other - so they run in line.)
        TIMED("k_seg_group", st, hipLaunchKernelGGL(k_seg_group<false>, dim3(persistent_grid(
                  reinterpret_cast<const void*>(k_seg_group<false>), 256, 0)), dim3(256), 0, st, fp_, fp_.cnt_vi, fp_.cnt_cm, &dh->cnt_n));
        TIMED("k_seg_group_heavy", st, hipLaunchKernelGGL(k_seg_group<true>, dim3(persistent_grid(
                  reinterpret_cast<const void*>(k_seg_group<true>), 256, 0)), dim3(256), 0, st, fp_, fp_.heavy_vi, fp_.heavy_cm, &dh->heavy_n));
        // second heavy pass: what the light kernel could not group (alphabets other than {A,C,G,T,N,-}; usually nothing)
        TIMED("k_seg_group_heavy2", st, hipLaunchKernelGGL(k_seg_group<true>, dim3(persistent_grid(
                  reinterpret_cast<const void*>(k_seg_group<true>), 256, 0)), dim3(256), 0, st, fp_, fp_.heavy2_vi, fp_.heavy2_cm, &dh->heavy2_n));
        sp.gcache_stride = gc_stride_; sp.gcache_cap = gc_region_ / gc_stride_;
        sp.list = fp_.slow_list; sp.list_n = fp_.slow_count; sp.gcache = gcache_.as<uint8_t>();
        TIMED("k_seg_count_slow", st, hipLaunchKernelGGL(k_seg_count<false>, dim3(seg_grid()), dim3(GT), seg_lds_, st, sp));
        sp.list = fp_.slow_list2; sp.list_n = fp_.slow_count2; sp.gcache = gcache_.as<uint8_t>() + gc_region_;
        TIMED("k_seg_count_slow2", st, hipLaunchKernelGGL(k_seg_count<false>, dim3(seg_grid()), dim3(GT), seg_lds_, st, sp));
    } else {
        sp.gcache_stride = gc_stride_; sp.gcache_cap = 2 * gc_region_ / gc_stride_; sp.gcache = gcache_.as<uint8_t>();
        if (big_) TIMED("k_seg_count", st, hipLaunchKernelGGL(k_seg_count<true>, dim3(seg_grid()), dim3(GT), seg_lds_, st, sp));
        else TIMED("k_seg_count", st, hipLaunchKernelGGL(k_seg_count<false>, dim3(seg_grid()), dim3(GT), seg_lds_, st, sp));
    }
    {   // text offsets of both outputs: one pass
        ScanSet<2> ss{{eds_len_.as<u64>(), seds_len_.as<u64>()}, {eds_len_.as<u64>(), seds_len_.as<u64>()}, {&dh->E, &dh->Q}};
        TIMED("scan_text", st, exclusive_scan_multi<2>(ss, d_nseg, scan_tmp_.as<u64>(), st));
    }
}

// workgroups that are resident at once: one persistent workgroup per slot
unsigned MsaPipeline::persistent_grid(const void* kern, int threads, size_t dyn_lds) const
{
    int per_cu = 1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, threads, dyn_lds) != hipSuccess || per_cu < 1)
        per_cu = 1;
    if (!cus_) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        const_cast<MsaPipeline*>(this)->cus_ = cus;
    }
    return (unsigned)(cus_ * per_cu);
}

unsigned MsaPipeline::seg_grid() const
{
    // persistent workgroups striding over the segments; several per CU to hide latency
    if (big_) return big_grid_;
    size_t per_cu = std::max<size_t>(1, std::min<size_t>(8, (150 * 1024) / std::max<size_t>(seg_lds_, 1)));
    return (unsigned)(256 * per_cu);
}

__global__ void k_copy_columns(MsaView mv, u64 col0, u64 ncols, uint8_t* __restrict__ out)
{
    const u64 total = (u64)mv.S * ncols;
    for (u64 t = blockIdx.x * (u64)blockDim.x + threadIdx.x; t < total; t += (u64)gridDim.x * blockDim.x) {
        const u64 r = t / ncols, c = col0 + t % ncols;
        out[t] = mv.file[mv.row_start[r] + mv.raw(c)];
    }
}

// first / last segment of the planned alignment: type, width and text sizes (for the slab stitch).  One small kernel
// gathers the ten numbers, one copy brings them over (the stitch runs inside every multi-GPU step).
__global__ void k_edge_info(const u64* __restrict__ seg_start, const u64* __restrict__ eds_off, const u64* __restrict__ seds_off,
                            const u64* __restrict__ V, u64 nseg, u64 L, u64* __restrict__ out)
{
    if (threadIdx.x || blockIdx.x) return;
    out[0] = seg_start[0]; out[1] = nseg > 1 ? seg_start[1] : L; out[2] = seg_start[nseg - 1];
    out[3] = nseg > 1 ? eds_off[1] : 0; out[4] = nseg > 1 ? seds_off[1] : 0;
    out[5] = eds_off[nseg - 1]; out[6] = seds_off[nseg - 1];
    out[7] = V[0] & 1ull; out[8] = (V[(L - 1) >> 6] >> ((L - 1) & 63)) & 1ull;
}

MsaPipeline::Edges MsaPipeline::edge_info(hipStream_t st)
{
    if (!planned_) throw ParamError("edge_info needs a planned alignment");
    const u64 nseg = h_.nseg;
    Edges e{};
    e.nseg = nseg;
    idx_tmp_.ensure(16 * sizeof(u64));
    u64 h[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    hipLaunchKernelGGL(k_edge_info, dim3(1), dim3(64), 0, st, seg_start_p_, eds_len_.as<u64>(), seds_len_.as<u64>(), mv_.V, nseg, h_.L,
                       idx_tmp_.as<u64>());
    EDSX_HIP(hipMemcpyAsync(h, idx_tmp_.ptr, sizeof(h), hipMemcpyDeviceToHost, st));
    EDSX_HIP(hipStreamSynchronize(st));
    e.fvar = h[7];
    e.fcols = h[1] - h[0];
    e.feds = nseg > 1 ? h[3] : h_.E;
    e.fseds = nseg > 1 ? h[4] : h_.Q;
    e.lvar = h[8];
    e.lcols = h_.L - h[2];
    e.leds = h_.E - h[5];
    e.lseds = h_.Q - h[6];
    return e;
}

// l-EDS stitch: the first and the last standalone common run of at least min_cols columns (msa_transforms.cpp:153) - the
// text between two such anchors does not depend on anything outside them
__global__ void k_anchor_find(const u64* __restrict__ seg_start, const u64* __restrict__ V, u64 nseg, u64 min_cols, u64* __restrict__ out)
{
    for (u64 seg = blockIdx.x * (u64)blockDim.x + threadIdx.x; seg < nseg; seg += (u64)gridDim.x * blockDim.x) {
        const u64 a = seg_start[seg], b = seg_start[seg + 1];
        const bool variant = (V[a >> 6] >> (a & 63)) & 1ull;
        if (!variant && b - a >= min_cols) {
            atomicMin((unsigned long long*)&out[0], (unsigned long long)seg);
            atomicMax((unsigned long long*)&out[1], (unsigned long long)(seg + 1));
        }
    }
}
__global__ void k_anchor_read(const u64* __restrict__ seg_start, const u64* __restrict__ eds_off, const u64* __restrict__ seds_off,
                              u64 nseg, u64 E, u64 Q, u64* __restrict__ out)
{
    if (threadIdx.x || blockIdx.x) return;
    const u64 first = out[0], last1 = out[1];
    if (last1 == 0) return;                                   // none
    const u64 last = last1 - 1;
    out[2] = seg_start[last]; out[3] = eds_off[last]; out[4] = seds_off[last];
    out[5] = seg_start[first + 1];
    out[6] = first + 1 < nseg ? eds_off[first + 1] : E;
    out[7] = first + 1 < nseg ? seds_off[first + 1] : Q;
}

MsaPipeline::Anchors MsaPipeline::anchor_info(u64 min_cols, hipStream_t st)
{
    if (!planned_) throw ParamError("anchor_info needs a planned alignment");
    Anchors a{};
    a.nseg = h_.nseg;
    idx_tmp_.ensure(16 * sizeof(u64));
    u64 h[8] = {~0ull, 0, 0, 0, 0, 0, 0, 0};
    EDSX_HIP(hipMemcpyAsync(idx_tmp_.ptr, h, sizeof(h), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_anchor_find, dim3(1024), dim3(256), 0, st, seg_start_p_, mv_.V, h_.nseg, min_cols, idx_tmp_.as<u64>());
    hipLaunchKernelGGL(k_anchor_read, dim3(1), dim3(64), 0, st, seg_start_p_, eds_len_.as<u64>(), seds_len_.as<u64>(), h_.nseg,
                       h_.E, h_.Q, idx_tmp_.as<u64>());
    EDSX_HIP(hipMemcpyAsync(h, idx_tmp_.ptr, sizeof(h), hipMemcpyDeviceToHost, st));
    EDSX_HIP(hipStreamSynchronize(st));
    if (h[1] == 0) return a;                                  // no such run
    a.found = 1;
    a.first_seg = h[0]; a.last_seg = h[1] - 1;
    a.last_col = h[2]; a.last_eds = h[3]; a.last_seds = h[4];
    a.first_end = h[5]; a.first_eds_end = h[6]; a.first_seds_end = h[7];
    return a;
}

// first segment that starts at or after alignment column `col` (binary search over the device table, one
// 8-byte copy per probe: a reporting / verification helper, not on the transform path)
MsaPipeline::SegLoc MsaPipeline::locate(u64 col, hipStream_t st)
{
    if (!planned_) throw ParamError("locate needs a planned alignment");
    const u64 nseg = h_.nseg;
    auto at = [&](const u64* arr, u64 i) -> u64 {
        u64 v = 0;
        EDSX_HIP(hipMemcpyAsync(&v, arr + i, 8, hipMemcpyDeviceToHost, st));
        EDSX_HIP(hipStreamSynchronize(st));
        return v;
    };
    u64 lo = 0, hi = nseg;                                  // seg_start[nseg] == L
    while (lo < hi) {
        const u64 mid = lo + (hi - lo) / 2;
        if (at(seg_start_p_, mid) >= col) hi = mid; else lo = mid + 1;
    }
    SegLoc r{};
    r.seg = lo;
    r.col = lo < nseg ? at(seg_start_p_, lo) : h_.L;
    r.eds_off = lo < nseg ? at(eds_len_.as<u64>(), lo) : h_.E;
    r.seds_off = lo < nseg ? at(seds_len_.as<u64>(), lo) : h_.Q;
    return r;
}

void MsaPipeline::copy_columns(u64 col0, u64 ncols, uint8_t* host_out, hipStream_t st)
{
    if (!planned_) throw ParamError("copy_columns needs a planned alignment");
    if (col0 + ncols > h_.L || ncols == 0) throw ParamError("copy_columns: column range outside the alignment");
    const size_t bytes = (size_t)h_.S * ncols;
    colbuf_.ensure(bytes + 16);
    hipLaunchKernelGGL(k_copy_columns, dim3(256), dim3(256), 0, st, mv_, col0, ncols, colbuf_.as<uint8_t>());
    EDSX_HIP(hipMemcpyAsync(host_out, colbuf_.ptr, bytes, hipMemcpyDeviceToHost, st));
    EDSX_HIP(hipStreamSynchronize(st));
}

void MsaPipeline::emit(uint8_t* d_eds, uint8_t* d_seds, hipStream_t st)
{
    if (!planned_) throw ParamError("edsx_msa_emit_device called without a successful plan");
    EmitParams ep;
    ep.mv = mv_; ep.seg_start = seg_start_p_; ep.nseg_ptr = nseg_p_; ep.Hseg = hseg_p_; ep.segbase = segbase_p_;
    ep.eds_off = eds_len_.as<u64>(); ep.seds_off = seds_len_.as<u64>(); ep.eds = d_eds; ep.seds = d_seds;
    ep.nwords = h_.nwords;
    ep.list = nullptr; ep.list_n = nullptr;
    ep.stage_cols = stage_cols_; ep.stage_off = stage_off_;
    ep.scratch = seg_scratch_.as<uint8_t>(); ep.scratch_stride = seg_scratch_stride_;
    EDSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_emit_variant<false>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)seg_lds_));
    auto launch_common = [&](hipStream_t s) {                 // the common segments: "{" reference text "}" and "{0}"
        TIMED("k_emit_common_seg", s, hipLaunchKernelGGL(k_emit_common_seg, dim3(4096), dim3(256), 0, s, mv_, seg_start_p_, nseg_p_,
                                                         eds_len_.as<u64>(), seds_len_.as<u64>(), d_eds, d_seds));
        TIMED("k_emit_common_long", s, hipLaunchKernelGGL(k_emit_common_long, dim3(512), dim3(256), 0, s, mv_, seg_start_p_,
                                                          eds_len_.as<u64>(), seds_len_.as<u64>(), long_list_.as<u64>(),
                                                          &hdr_.as<MsaHdr>()->long_n, d_eds, d_seds));
    };
    if (fast_) {
        FastParams fp = fp_;
        fp.eds = d_eds; fp.seds = d_seds;
        auto launch_emit = [&](auto kern, const char* name) {
            TIMED(name, st, hipLaunchKernelGGL(kern, dim3(persistent_grid(reinterpret_cast<const void*>(kern), 256, 0)),
                                               dim3(256), 0, st, fp));
        };
        // The main emitter fills the machine (beside it the small emitters only slow it down: measured in rounds 1 and 3);
        // behind it the wide emitter, the generic emitter (a few hundred workgroup-sized segments) and the common text
        // run side by side on three streams.
        launch_emit(k_emit_fast2, "k_emit_fast");
        ensure_side_streams();
        hipStream_t s1 = side_[0], s2 = side_[1];
#if defined(EDSX_EXPERIMENTS) && defined(EDSX_TAIL_SERIAL)
        s1 = st; s2 = st;                                   // every kernel alone: what each costs by itself
#endif
        EDSX_HIP(hipEventRecord(side_ev_[0], st));
        EDSX_HIP(hipStreamWaitEvent(s1, side_ev_[0], 0));
        EDSX_HIP(hipStreamWaitEvent(s2, side_ev_[0], 0));
        {
            auto launch_wide = [&](auto kern, const char* name) {
                TIMED(name, st, hipLaunchKernelGGL(kern, dim3(persistent_grid(reinterpret_cast<const void*>(kern), 256, 0)),
                                                   dim3(256), 0, st, fp));
            };
            if (h_.S >= 1000) {                                          // ids of five bytes exist
                launch_wide(k_emit_fast<true, true, 8>, "k_emit_fast_wide8"); launch_wide(k_emit_fast<true, true, 16>, "k_emit_fast_wide16");
            } else {
                launch_wide(k_emit_fast<false, true, 8>, "k_emit_fast_wide8"); launch_wide(k_emit_fast<false, true, 16>, "k_emit_fast_wide16");
            }
        }
        ep.gcache_stride = gc_stride_; ep.gcache_cap = gc_region_ / gc_stride_;
        ep.list = fp_.slow_list; ep.list_n = fp_.slow_count; ep.gcache = gcache_.as<uint8_t>();
        ep.list2 = fp_.slow_list2; ep.list2_n = fp_.slow_count2; ep.gcache2 = gcache_.as<uint8_t>() + gc_region_;
        TIMED("k_emit_variant_slow", s2, hipLaunchKernelGGL(k_emit_variant<false>, dim3(seg_grid()), dim3(GT), seg_lds_, s2, ep));
        launch_common(s1);
        EDSX_HIP(hipEventRecord(side_ev_[1], s1));
        EDSX_HIP(hipEventRecord(side_ev_[2], s2));
        EDSX_HIP(hipStreamWaitEvent(st, side_ev_[1], 0));
        EDSX_HIP(hipStreamWaitEvent(st, side_ev_[2], 0));
    } else {
        launch_common(st);
        ep.gcache_stride = gc_stride_; ep.gcache_cap = 2 * gc_region_ / gc_stride_; ep.gcache = gcache_.as<uint8_t>();
        if (big_) TIMED("k_emit_variant", st, hipLaunchKernelGGL(k_emit_variant<true>, dim3(seg_grid()), dim3(GT), seg_lds_, st, ep));
        else TIMED("k_emit_variant", st, hipLaunchKernelGGL(k_emit_variant<false>, dim3(seg_grid()), dim3(GT), seg_lds_, st, ep));
    }
    EDSX_HIP(hipGetLastError());
}

} // namespace edsx
