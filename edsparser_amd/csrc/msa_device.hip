// msa_device.hip — MSA -> EDS / l-EDS on gfx950 (MI355X).  Hand-written HIP, wave64, no MFMA:
// this is byte/bit work bounded by HBM bandwidth.
//
// Replaces the reference's three passes (src/cpp/lib/transforms/msa_transforms.cpp):
//   pass 1 parse_msa_and_build_variant_bv :36-90   -> k_find_*/k_index_rows + k_scan_extract
//   pass 2 build_eds/leds_boundaries      :101-190 -> k_runstart_words .. k_write_segs
//   pass 3 generate_output                :200-324 -> k_seg_count, scans, k_emit_common, k_emit_variant
//
// Data layout in HBM
//   file image    the FASTA bytes as given (no repacking).  Row r's raw byte q lives at
//                 row_start[r] + q; raw position q of alignment column c is c + c/lw for wrapped
//                 rows (msa_transforms.cpp:268-269) and c for one-line rows.
//   Vraw / V      1 bit per raw position / alignment column: 1 = variant column (some row differs
//                 from row 0, or row 0 has '-').  Complement of the reference's bit-vector B.
//   vc            the variant columns only, column-major: vc[slot * Spad + r] = byte of row r.
//                 Slots are handed out per tile by an atomic counter (unordered between tiles,
//                 consecutive inside a 64-column word); word_slot[w] = slot of the first variant
//                 column of raw word w.  Every later kernel reads rows through vc, so the S x L
//                 matrix is read from HBM exactly once.
//   run/seg table seg_start[nseg+1] (alignment columns), bitmap Hseg of segment starts with a
//                 per-word prefix segbase[]; a segment is common iff V[seg_start] == 0.
//   eds_off/seds_off  exclusive scans of the per-segment text sizes.
#include "msa_device.hpp"

#include <algorithm>
#include <cstring>

namespace edsx {

// ---------------------------------------------------------------------------------------------
// device header block
// ---------------------------------------------------------------------------------------------
enum : u64 {
    ST_NOT_FASTA = 1, ST_LAYOUT = 2, ST_TOO_MANY_ROWS = 4, ST_VC_OVERFLOW = 8,
    ST_NEWLINE_IN_DATA = 16, ST_FEW_ROWS = 32
};

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
    u64 p = 0, s = 0, bad = 0;
    while (true) {
        if (f[p] != '>') {
            // tolerate blank lines at the very end (skipped by the reference, :47-49)
            u64 rest = n - p;
            if (rest > 4096) { bad = ST_LAYOUT; break; }
            u64 nonnl = 0;
            for (u64 i = lane; i < rest; i += 64) nonnl |= (f[p + i] != '\n');
            if (ballot64(nonnl != 0)) bad = ST_LAYOUT;
            break;
        }
        u64 nlpos = n;
        for (u64 base = p; base < n; base += 64) {
            u64 i = base + lane;
            u64 b = ballot64(i < n && f[i] == '\n');
            if (b) { nlpos = base + __builtin_ctzll(b); break; }
        }
        if (nlpos >= n) { bad = ST_LAYOUT; break; }
        u64 st = nlpos + 1;
        if (st + Draw > n) { bad = ST_LAYOUT; break; }
        if (s >= row_cap) { bad = ST_TOO_MANY_ROWS; break; }
        if (lane == 0) row_start[s] = st;
        s++;
        u64 q = st + Draw;
        if (q == n) break;                             // no trailing newline
        if (f[q] != '\n') { bad = ST_LAYOUT; break; }
        p = q + 1;
        if (p == n) break;
    }
    if (lane == 0) {
        if (s < 2 && !bad) bad = ST_FEW_ROWS;
        h->status |= bad;
        h->S = s; h->L = L; h->lw = wrapped ? lw : 0; h->Draw = Draw;
        h->nwords = (L + 63) / 64;
        h->nwords_raw = (Draw + 63) / 64;
    }
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
};

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

template <int T, int RPT, bool HOLD>
__global__ void __launch_bounds__(T) k_scan_extract(K1Params p)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t colbuf[];
    __shared__ u32 D[256];
    __shared__ u32 pre[256];
    __shared__ u32 wtot[4];
    __shared__ u64 slot_base_sh;
    __shared__ u32 nv_sh;

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
    u64* rs = reinterpret_cast<u64*>(colbuf);
    for (u32 r = tid; r < p.S; r += T) rs[r] = p.row_start[r];
    if (tid < 256) D[tid] = 0;
    __syncthreads();

    const uint8_t* f = p.file;
    const u32 Sm1 = p.S - 1;
    uint4 ref = make_uint4(0, 0, 0, 0);
    uint4 d[HOLD ? RPT : 1];
    uint4 acc = make_uint4(0, 0, 0, 0);                        // OR over rows of (row ^ ref): a byte is
    if (full_tile) {                                           // non-zero iff some row differs there                                           // fast path: unconditional 16-B loads
        ref = load16u(f + rs[0] + q);
        uint4 v[RPT];
#pragma unroll
        for (int it = 0; it < RPT; it++) {
            const u32 r = sub + it * RI;
            v[it] = load16u(f + rs[r < p.S ? r : Sm1] + q);    // clamped: rows past S re-read row S-1
        }
#pragma unroll
        for (int it = 0; it < RPT; it++) {
            if constexpr (HOLD) d[it] = v[it];
            acc.x |= v[it].x ^ ref.x; acc.y |= v[it].y ^ ref.y;   // a clamped duplicate changes nothing
            acc.z |= v[it].z ^ ref.z; acc.w |= v[it].w ^ ref.w;
        }
    } else {
        if (nb > 0) ref = load_partial(f + rs[0] + q, nb);
#pragma unroll
        for (int it = 0; it < RPT; it++) {
            const u32 r = sub + it * RI;
            uint4 v = ref;
            if (r < p.S && nb > 0) v = load_partial(f + rs[r] + q, nb);
            if constexpr (HOLD) d[it] = v;
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
    if (diff) atomicOr(&D[j], diff);
    __syncthreads();

    const u32 V16 = D[j];
    if (V16 & nlmask) bad = 1;                                 // a row deviates at a newline slot

    // exclusive prefix of popc(D[*]) over the tile's chunks (cpr <= 256 -> <= 4 waves)
    if (tid < 256) {
        u32 c = tid < cpr ? __builtin_popcount(D[tid]) : 0;
        u32 incl = c;
        for (int o = 1; o < 64; o <<= 1) { u32 a = __shfl_up(incl, o, 64); if ((tid & 63) >= (u32)o) incl += a; }
        if ((tid & 63) == 63) wtot[tid >> 6] = incl;
        pre[tid] = incl - c;
    }
    __syncthreads();
    if (tid < 256) {
        u32 off = 0;
        for (u32 w = 0; w < (tid >> 6); w++) off += wtot[w];
        pre[tid] += off;
    }
    if (tid == 0) {
        u32 nv = wtot[0] + wtot[1] + wtot[2] + wtot[3];
        nv_sh = nv;
        u64 base = nv ? atomicAdd(&p.hdr->nv, (u64)nv) : 0;
        slot_base_sh = base;
        if (base + nv > p.vc_cap_cols) atomicOr(&p.hdr->status, (u64)ST_VC_OVERFLOW);
    }
    __syncthreads();
    const u64 slot_base = slot_base_sh;
    const u32 nv = nv_sh;
    const bool overflow = slot_base + nv > p.vc_cap_cols;

    if (tid < cpr / 4) {                                       // V words + per-word slot base
        u64 wi = q0 / 64 + tid;
        if (wi * 64 < p.Draw) {
            u64 bits = (u64)D[4 * tid] | ((u64)D[4 * tid + 1] << 16) | ((u64)D[4 * tid + 2] << 32) |
                       ((u64)D[4 * tid + 3] << 48);
            p.Vraw[wi] = bits;
            p.word_slot[wi] = slot_base + pre[4 * tid];
        }
    }

    // extraction: variant bytes -> LDS (column-major) -> HBM, in batches of cap_cols columns
    if (!overflow && nv) {
        const u32 cap = p.cap_cols;
        for (u32 b0 = 0; b0 < nv; b0 += cap) {
            if (V16) {
                u32 idx = pre[j];
                if constexpr (HOLD) {
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
            __syncthreads();
            const u32 ncols = (nv - b0) < cap ? (nv - b0) : cap;
            const size_t nbytes = (size_t)ncols * p.Spad;      // Spad % 16 == 0
            uint8_t* g = p.vc + (slot_base + b0) * (u64)p.Spad;
            for (size_t o = (size_t)tid * 16; o < nbytes; o += (size_t)T * 16)
                *reinterpret_cast<uint4*>(g + o) = *reinterpret_cast<const uint4*>(colbuf + o);
            __syncthreads();
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
constexpr int GT = 256;
constexpr u32 GID_NONE = 0xffffu;

struct SegLds {
    u64* key; u32* rep_row; u32* run; uint16_t* gid;
    __device__ SegLds(uint8_t* base, u32 S)
    {
        key = reinterpret_cast<u64*>(base);
        rep_row = reinterpret_cast<u32*>(base + (size_t)8 * S);
        run = reinterpret_cast<u32*>(base + (size_t)12 * S);
        gid = reinterpret_cast<uint16_t*>(base + (size_t)16 * S);
    }
};

__device__ __forceinline__ u32 seg_col_byte(const MsaView& mv, u64 c, u32 r, u32 isvar, u64 slot)
{
    return isvar ? mv.vc[slot * mv.Spad + r] : mv.ref_byte(c);
}

// gap-stripped string of row r over [a,b): the reference drops '\n' and '-' and stops at '\0'
// (msa_transforms.cpp:281-286)
__device__ u64 seg_row_key(const MsaView& mv, u64 a, u64 b, u32 r, bool exact)
{
    u64 key = exact ? 0ull : 0xcbf29ce484222325ull;
    u32 len = 0;
    for (u64 c = a; c < b; c++) {
        u32 isvar = mv.vbit(c);
        u64 sl = isvar ? mv.slot(c) : 0;
        u32 ch = seg_col_byte(mv, c, r, isvar, sl);
        if (ch == 0) break;
        if (ch == '-' || ch == '\n') continue;
        if (exact) key |= (u64)ch << (8 * len);
        else key = (key ^ ch) * 0x100000001b3ull;
        len++;
    }
    if (!exact) key = (key ^ len) * 0x100000001b3ull;
    return key;
}

__device__ bool seg_rows_equal(const MsaView& mv, u64 a, u64 b, u32 r1, u32 r2)
{
    u64 c1 = a, c2 = a;
    while (true) {
        u32 x = 0, y = 0;
        while (c1 < b) {
            u32 isvar = mv.vbit(c1);
            u32 ch = seg_col_byte(mv, c1, r1, isvar, isvar ? mv.slot(c1) : 0);
            if (ch == 0) { c1 = b; break; }
            c1++;
            if (ch != '-' && ch != '\n') { x = ch; break; }
        }
        while (c2 < b) {
            u32 isvar = mv.vbit(c2);
            u32 ch = seg_col_byte(mv, c2, r2, isvar, isvar ? mv.slot(c2) : 0);
            if (ch == 0) { c2 = b; break; }
            c2++;
            if (ch != '-' && ch != '\n') { y = ch; break; }
        }
        if (x != y) return false;
        if (x == 0) return true;          // both exhausted
    }
}

__device__ u32 seg_row_len(const MsaView& mv, u64 a, u64 b, u32 r)
{
    u32 len = 0;
    for (u64 c = a; c < b; c++) {
        u32 isvar = mv.vbit(c);
        u32 ch = seg_col_byte(mv, c, r, isvar, isvar ? mv.slot(c) : 0);
        if (ch == 0) break;
        if (ch != '-' && ch != '\n') len++;
    }
    return len;
}

// returns k (number of distinct strings); fills lds.gid[], lds.rep_row[0..k)
__device__ u32 group_segment(const MsaView& mv, u64 a, u64 b, SegLds& lds, u32* rep_sh)
{
    const u32 S = mv.S;
    const bool exact = (b - a) <= 8;
    for (u32 r = threadIdx.x; r < S; r += GT) {
        lds.key[r] = seg_row_key(mv, a, b, r, exact);
        lds.gid[r] = GID_NONE;
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
                (exact || r == rep || seg_rows_equal(mv, a, b, r, rep)))
                lds.gid[r] = (uint16_t)g;
        }
        if (threadIdx.x == 0) lds.rep_row[g] = rep;
        g++;
        __syncthreads();
    }
    return g;
}

struct SegParams {
    MsaView mv; const u64* seg_start; const u64* nseg_ptr; u64* eds_len; u64* seds_len; u64 tok_total;
};

// K3: per-segment output sizes.  common: "{" ref "}" and "{0}"; variant: see generate_output.
__global__ void __launch_bounds__(GT) k_seg_count(SegParams p)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    __shared__ u32 rep_sh;
    __shared__ u64 sum_sh;
    SegLds lds(lds_raw, p.mv.S);
    if (p.mv.hdr->status) return;                     // vc overflow: the host grows vc and replans
    const u64 nseg = *p.nseg_ptr;
    for (u64 seg = blockIdx.x; seg < nseg; seg += gridDim.x) {
        const u64 a = p.seg_start[seg], b = p.seg_start[seg + 1];
        if (!p.mv.vbit(a)) {
            if (threadIdx.x == 0) { p.eds_len[seg] = 2 + (b - a); p.seds_len[seg] = 3; }
            continue;
        }
        if (threadIdx.x == 0) sum_sh = 0;
        const u32 k = group_segment(p.mv, a, b, lds, &rep_sh);
        u64 mine = 0;
        for (u32 g = threadIdx.x; g < k; g += GT) mine += seg_row_len(p.mv, a, b, lds.rep_row[g]);
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
// K5a: common-segment text.  One wave per 64-column word, lane = column.
//   msa_transforms.cpp:245-258
// ---------------------------------------------------------------------------------------------
struct EmitParams {
    MsaView mv; const u64* seg_start; const u64* nseg_ptr; const u64* Hseg; const u64* segbase;
    const u64* eds_off; const u64* seds_off; uint8_t* eds; uint8_t* seds; u64 nwords;
};

__global__ void __launch_bounds__(256) k_emit_common(EmitParams p)
{
    const u32 lane = threadIdx.x & 63;
    const u64 wave = (blockIdx.x * (u64)blockDim.x + threadIdx.x) >> 6;
    const u64 nwaves = ((u64)gridDim.x * blockDim.x) >> 6;
    for (u64 w = wave; w < p.nwords; w += nwaves) {
        const u64 c = w * 64 + lane;
        if (c >= p.mv.L) continue;
        const u64 hbits = p.Hseg[w];
        const u64 below = hbits & ((2ull << lane) - 1ull);
        const u64 seg = p.segbase[w] + __builtin_popcountll(below) - 1;
        const u64 a = p.seg_start[seg];
        if (p.mv.vbit(a)) continue;                          // variant segment: k_emit_variant
        const u64 e = p.seg_start[seg + 1];
        const u64 pos = p.eds_off[seg] + 1 + (c - a);
        p.eds[pos] = (uint8_t)p.mv.ref_byte(c);
        if (c == a) {
            p.eds[pos - 1] = '{';
            uint8_t* s = p.seds + p.seds_off[seg];
            s[0] = '{'; s[1] = '0'; s[2] = '}';
        }
        if (c == e - 1) p.eds[pos + 1] = '}';
    }
}

// ---------------------------------------------------------------------------------------------
// K5b: variant-segment text.  msa_transforms.cpp:297-317.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void write_decimal(uint8_t* dst, u32 v, u32 nd)
{
    for (int i = (int)nd - 1; i >= 0; i--) { dst[i] = (uint8_t)('0' + v % 10u); v /= 10u; }
}

__global__ void __launch_bounds__(GT) k_emit_variant(EmitParams p)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    __shared__ u32 rep_sh;
    const MsaView& mv = p.mv;
    const u32 S = mv.S;
    SegLds lds(lds_raw, S);
    if (mv.hdr->status) return;
    const u64 nseg = *p.nseg_ptr;
    const u32 lane = threadIdx.x & 63;
    for (u64 seg = blockIdx.x; seg < nseg; seg += gridDim.x) {
        const u64 a = p.seg_start[seg], b = p.seg_start[seg + 1];
        if (!mv.vbit(a)) continue;
        const u32 k = group_segment(mv, a, b, lds, &rep_sh);
        uint8_t* eds = p.eds + p.eds_off[seg];
        uint8_t* seds = p.seds + p.seds_off[seg];

        // ---- eds: "{" s0 "," s1 ... "}" ; key[] is reused for the string offsets
        u64* goff = lds.key;
        for (u32 g = threadIdx.x; g < k; g += GT) {
            goff[g] = seg_row_len(mv, a, b, lds.rep_row[g]);
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
                u32 isvar = mv.vbit(c);
                u32 ch = seg_col_byte(mv, c, r, isvar, isvar ? mv.slot(c) : 0);
                if (ch == 0) break;
                if (ch != '-' && ch != '\n') *dst++ = (uint8_t)ch;
            }
            *dst = (g + 1 < k) ? ',' : '}';
            seds[lds.run[g] - 1] = '{';
        }
        __syncthreads();

        // ---- seds: wave 0 walks the rows in order; run[g] = next write offset of group g
        if (threadIdx.x < 64) {
            for (u32 base = 0; base < S; base += 64) {
                const u32 r = base + lane;
                const bool valid = r < S;
                const u32 g = valid ? lds.gid[r] : 0xffffffffu;
                const u32 tl = ndigits(r + 1) + 1;
                const u32 tlA = __shfl(tl, 0, 64);              // <= 2 token lengths per 64 rows
                const u64 maskA = ballot64(valid && tl == tlA);
                u64 todo = ballot64(valid);
                while (todo) {
                    const int leader = __builtin_ctzll(todo);
                    const u32 g0 = __shfl(g, leader, 64);
                    const u64 m = ballot64(valid && g == g0);
                    const u32 start = lds.run[g0];
                    if (valid && g == g0) {
                        const u32 preA = mbcnt(m & maskA), preB = mbcnt(m & ~maskA);
                        uint8_t* dst = seds + start + preA * tlA + preB * (tlA + 1);
                        write_decimal(dst, r + 1, tl - 1);
                        dst[tl - 1] = ',';
                    }
                    const u32 tot = __builtin_popcountll(m & maskA) * tlA +
                                    __builtin_popcountll(m & ~maskA) * (tlA + 1);
                    if (lane == (u32)leader) lds.run[g0] = start + tot;
                    todo &= ~m;
                }
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
    if (!timed_.empty()) (void)hipEventSynchronize(timed_.back().t1);
    for (auto& t : timed_) {
        float v = 0;
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

MsaPipeline::~MsaPipeline() { clear_timers(); }

static const char* status_message(u64 st)
{
    if (st & ST_NOT_FASTA) return "Invalid MSA: expected a FASTA header line starting with '>'";
    if (st & ST_FEW_ROWS) return "Invalid MSA: at least two sequences are required";
    if (st & ST_TOO_MANY_ROWS) return "MSA has more sequences than this build supports (8192)";
    if (st & ST_NEWLINE_IN_DATA) return "Invalid MSA: rows must have equal length and a uniform line width";
    if (st & ST_LAYOUT) return "Invalid MSA: rows must have equal length and a uniform line width";
    return "MSA transform failed";
}

template <int T, int RPT, bool HOLD>
static void launch_k1(const K1Params& p, size_t lds, hipStream_t st)
{
    auto kern = k_scan_extract<T, RPT, HOLD>;
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
    rows_.ensure(sizeof(u64) * ROW_CAP);
    MsaHdr* dh = hdr_.as<MsaHdr>();
    EDSX_HIP(hipMemsetAsync(dh, 0, sizeof(MsaHdr), st));

    // ---- K0: geometry + row index (one host sync: everything below is sized from it)
    TIMED("k_find_hdr_end", st, hipLaunchKernelGGL(k_find_hdr_end, dim3(1), dim3(64), 0, st, d_msa, (u64)n, dh));
    TIMED("k_find_row0", st, hipLaunchKernelGGL(k_find_row0, dim3(512), dim3(1024), 0, st, d_msa, (u64)n, dh));
    TIMED("k_index_rows", st, hipLaunchKernelGGL(k_index_rows, dim3(1), dim3(64), 0, st, d_msa, (u64)n, dh,
                                                 rows_.as<u64>(), (u64)ROW_CAP));
    EDSX_HIP(hipMemcpyAsync(&h_, dh, sizeof(MsaHdr), hipMemcpyDeviceToHost, st));
    EDSX_HIP(hipStreamSynchronize(st));
    if (h_.status) throw FormatError(status_message(h_.status));
    if (h_.S > MAX_ROWS) throw FormatError(status_message(ST_TOO_MANY_ROWS));

    const u64 Draw = h_.Draw;
    const u32 Spad = (u32)((h_.S + 15) / 16 * 16);
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
    const u32 Spad = (u32)((S + 15) / 16 * 16);
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
    scan_tmp_.ensure(8 * ((L + 2) / SCAN_TILE + 2));
    run_start_.ensure(8 * (L + 2));
    if (l) { flag_.ensure(8 * (L + 2)); seg_start_.ensure(8 * (L + 2)); }
    eds_len_.ensure(8 * (L + 2));
    seds_len_.ensure(8 * (L + 2));

    // ---- K1 geometry: largest tile whose rows fit the per-thread register budget
    constexpr int T = 1024, RPT = 16;
    u32 cpr_log2 = 8;
    while (cpr_log2 > 2 && (u64)RPT * (T >> cpr_log2) < S) cpr_log2--;
    const bool hold = (u64)RPT * (T >> cpr_log2) >= S;
    const u64 W = 16ull << cpr_log2;
    const u64 ntiles = (Draw + W - 1) / W;
    const size_t colbuf_bytes = 96 * 1024;
    if (ntiles > 0x7fffffffull) throw FormatError("MSA too large for one launch");

    K1Params kp;
    kp.file = d_msa; kp.row_start = rows_.as<u64>(); kp.hdr = dh;
    kp.Vraw = vraw_.as<u64>(); kp.word_slot = wslot_.as<u64>(); kp.vc = vc_.as<uint8_t>();
    kp.vc_cap_cols = vc_cap_cols_; kp.Draw = Draw; kp.lw = lw; kp.S = (u32)S; kp.Spad = Spad;
    kp.cpr_log2 = cpr_log2; kp.cap_cols = (u32)(colbuf_bytes / Spad); kp.ntiles = ntiles;
    launch_timer_begin("k_scan_extract", st);
    if (hold) launch_k1<T, RPT, true>(kp, colbuf_bytes, st);
    else launch_k1<T, RPT, false>(kp, colbuf_bytes, st);
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
    mv_.S = (u32)S; mv_.Spad = Spad;
    seg_start_p_ = seg_start; hseg_p_ = Hseg; segbase_p_ = segbase; nseg_p_ = d_nseg;
    seg_lds_ = (size_t)18 * S + 64;

    SegParams sp;
    sp.mv = mv_; sp.seg_start = seg_start; sp.nseg_ptr = d_nseg; sp.eds_len = eds_len_.as<u64>();
    sp.seds_len = seds_len_.as<u64>(); sp.tok_total = token_total((u32)S);
    EDSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_seg_count),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)seg_lds_));
    TIMED("k_seg_count", st, hipLaunchKernelGGL(k_seg_count, dim3(seg_grid()), dim3(GT), seg_lds_, st, sp));
    TIMED("scan_eds", st, exclusive_scan_u64(eds_len_.as<u64>(), eds_len_.as<u64>(), d_nseg, &dh->E,
                                             scan_tmp_.as<u64>(), st));
    TIMED("scan_seds", st, exclusive_scan_u64(seds_len_.as<u64>(), seds_len_.as<u64>(), d_nseg, &dh->Q,
                                              scan_tmp_.as<u64>(), st));
}

unsigned MsaPipeline::seg_grid() const
{
    // persistent workgroups striding over the segments; several per CU to hide latency
    size_t per_cu = std::max<size_t>(1, std::min<size_t>(8, (150 * 1024) / std::max<size_t>(seg_lds_, 1)));
    return (unsigned)(256 * per_cu);
}

void MsaPipeline::emit(uint8_t* d_eds, uint8_t* d_seds, hipStream_t st)
{
    if (!planned_) throw ParamError("edsx_msa_emit_device called without a successful plan");
    EmitParams ep;
    ep.mv = mv_; ep.seg_start = seg_start_p_; ep.nseg_ptr = nseg_p_; ep.Hseg = hseg_p_; ep.segbase = segbase_p_;
    ep.eds_off = eds_len_.as<u64>(); ep.seds_off = seds_len_.as<u64>(); ep.eds = d_eds; ep.seds = d_seds;
    ep.nwords = h_.nwords;
    TIMED("k_emit_common", st, hipLaunchKernelGGL(k_emit_common, dim3(2048), dim3(256), 0, st, ep));
    EDSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_emit_variant),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)seg_lds_));
    TIMED("k_emit_variant", st, hipLaunchKernelGGL(k_emit_variant, dim3(seg_grid()), dim3(GT), seg_lds_, st, ep));
    EDSX_HIP(hipGetLastError());
}

} // namespace edsx
