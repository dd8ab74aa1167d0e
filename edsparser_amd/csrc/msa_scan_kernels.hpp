// msa_scan_kernels.hpp - row index (K0), column scan + fused grouping (K1), runs -> segments (K2)
// included by msa_scan.hip, the translation unit of these kernels (msa_scan_launch.hpp is what the host side sees).
#pragma once
#include "msa_wave.hpp"
#include "msa_scan_launch.hpp"

namespace edsx {

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

constexpr u32 FUSE_MAXW = 10;      // widest run grouped by the column scan (exact 3-bit-per-column keys in one dword; two-dword keys
                                   // for 11..20 columns spill 71 registers here - rounds 2 and 3)
#ifndef EDSX_TAIL_WAVES
#define EDSX_TAIL_WAVES 8
#endif
constexpr u32 TAIL_WAVES = EDSX_TAIL_WAVES;   // waves of a scan workgroup that copy / group its variant columns (the rest retire early)
constexpr u32 CLIST = 2048;        // variant columns per tile in fused mode (the LDS image holds at most 64 KB / 32 B columns)



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
                uint8_t* const dst0 = p.vc + (slot_base + pre_of(j)) * (u64)p.Spad;
                for (u32 r = sub; r < p.S; r += 4 * RI) {          // rows outside, four in flight (see the batched path below)
                    uint4 v[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const u32 rr = r + (u32)u * RI;
                        v[u] = make_uint4(0, 0, 0, 0);
                        if (rr < p.S) v[u] = nb == 16 ? load16u(f + p.row_start[rr] + q) : load_partial(f + p.row_start[rr] + q, nb);
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const u32 rr = r + (u32)u * RI;
                        if (rr < p.S) {
                            u32 m = V16;
                            uint8_t* dst = dst0 + rr;
                            while (m) {
                                const int i = __builtin_ctz(m);
                                m &= m - 1;
                                const u32 ch = byte_of(v[u], i);
                                if (ch == '\n') bad = 1;
                                *dst = (uint8_t)ch;
                                dst += p.Spad;
                            }
                        }
                    }
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
                    // rows outside: a row's 16-byte chunk is fetched once more (it is in L2 from the first pass) and its
                    // variant bytes are picked out of the registers, four rows in flight - not a byte load per variant
                    // column and row behind a row-start load each (3000 rows x 1 M columns: scan 4.6 -> see DESIGN)
                    for (u32 r = sub; r < p.S; r += 4 * RI) {
                        uint4 v[4];
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            const u32 rr = r + (u32)u * RI;
                            v[u] = make_uint4(0, 0, 0, 0);
                            if (rr < p.S) v[u] = nb == 16 ? load16u(f + p.row_start[rr] + q) : load_partial(f + p.row_start[rr] + q, nb);
                        }
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            const u32 rr = r + (u32)u * RI;
                            if (rr < p.S) {
                                u32 m = V16, id = idx;
                                while (m) {
                                    const int i = __builtin_ctz(m);
                                    m &= m - 1;
                                    if (id >= b0 && id < b0 + cap) {
                                        const u32 ch = byte_of(v[u], i);
                                        if (ch == '\n') bad = 1;
                                        colbuf[(size_t)(id - b0) * p.Spad + rr] = (uint8_t)ch;
                                    }
                                    id++;
                                }
                            }
                        }
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

} // namespace edsx
