// msa_device.hip — MSA -> EDS / l-EDS on gfx950 (MI355X).  Hand-written HIP, wave64, no MFMA:
// this is byte/bit work bounded by HBM bandwidth.
//
// Replaces the reference's three passes (src/cpp/lib/transforms/msa_transforms.cpp):
//   pass 1 parse_msa_and_build_variant_bv :36-90   -> k_find_*, k_index_spec/_check/_rows + k_scan_extract
//   pass 2 build_eds/leds_boundaries      :101-190 -> k_runstart_words .. k_write_segs
//   pass 3 generate_output                :200-324 -> k_seg_meta, k_seg_count_fast (wave per segment, S <= 1024)
//                                                     / k_seg_count (workgroup per segment), scans,
//                                                     k_emit_common, k_emit_variant_fast / k_emit_variant
//
// Data layout in HBM
//   file image    the FASTA bytes as given (no repacking).  Row r's raw byte q lives at
//                 row_start[r] + q; raw position q of alignment column c is c + c/lw for wrapped
//                 rows (msa_transforms.cpp:268-269) and c for one-line rows.
//   Vraw / V      1 bit per raw position / alignment column: 1 = variant column (some row differs
//                 from row 0, or row 0 has '-').  Complement of the reference's bit-vector B.
//   vc            the variant columns only, column-major: vc[slot * Spad + vc_pos(r)] = byte of row r
//                 (rows permuted so that a lane's 16-byte load holds rows lane, lane+64, ...).
//                 Slots are handed out per tile by an atomic counter (unordered between tiles,
//                 consecutive inside a 64-column word); word_slot[w] = slot of the first variant
//                 column of raw word w.  Every later kernel reads rows through vc, so the S x L
//                 matrix is read from HBM exactly once.
//   run/seg table seg_start[nseg+1] (alignment columns), bitmap Hseg of segment starts with a
//                 per-word prefix segbase[]; a segment is common iff V[seg_start] == 0.
//   grec          one grouping record per variant segment (count -> emit): group id of every row (one
//                 byte, lane-major like vc), representative row and letter of every group, k.
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

// ---------------------------------------------------------------------------------------------
// K1: column scan + variant-column extraction.  One workgroup owns a tile of W = 16*CPR raw
// columns for ALL rows: T threads, thread (sub, j) holds the 16-byte chunk j of rows
// sub, sub+RI, ... (RI = T/CPR) in registers, so each input byte is read from HBM once.
//   msa_transforms.cpp:71-79  B[i] = 0 if c != ref[i] || c == '-'
// ---------------------------------------------------------------------------------------------
struct K1Params {
    const uint8_t* file; const u64* row_start; MsaHdr* hdr;
    u64* Vraw; u64* word_slot; uint8_t* vc; u64 vc_cap_cols;
    u64 Draw, lw; u32 S, Spad, Gp, cpr_log2, cap_cols /* LDS colbuf capacity in columns */;
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

template <int T, int RPT, bool HOLD, bool LANEROWS, int MINW>
__global__ void __launch_bounds__(T, MINW) k_scan_extract(K1Params p)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t colbuf[];
    __shared__ u32 D[256];
    __shared__ u32 pre[256];
    __shared__ u32 wtot[4];
    __shared__ u64 slot_base_sh;

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
    u64* rs = reinterpret_cast<u64*>(colbuf);
    for (u32 r = tid; r < p.S; r += T) rs[r] = p.row_start[r];
    if (tid < 256) D[tid] = 0;
    __syncthreads();

    const uint8_t* f = p.file;
    const u32 Sm1 = p.S - 1;
    // rows of this thread.  Plain: sub, sub + RI, ...  Lane rows (S <= 1024, RI == 4 * Gp): the rows
    // whose bytes are consecutive in a vc column, vc_pos(row_of(it)) == sub * 16 + it:
    //   row_of(it) = sub * (16 / Gp) + it / Gp + 64 * (it % Gp)      (Gp = 16: sub + 64 * it)
    const u32 gl = 31u - (u32)__builtin_clz(p.Gp);             // log2(Gp), Gp a power of two
    auto row_of = [&](u32 it) -> u32 {
        if constexpr (LANEROWS) return (sub << (4u - gl)) + (it >> gl) + 64u * (it & (p.Gp - 1u));
        else return sub + it * RI;
    };
    uint4 ref = make_uint4(0, 0, 0, 0);
    uint4 d[HOLD ? RPT : 1];
    uint4 acc = make_uint4(0, 0, 0, 0);                        // OR over rows of (row ^ ref): a byte is
    if (full_tile) {                                           // non-zero iff some row differs there
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
    // every thread completes its own prefix (no third barrier); the slot atomic is issued now and its
    // result is first needed after the extraction into LDS, which hides its ~1 us round trip
    const u32 w0 = wtot[0], w1 = wtot[1], w2 = wtot[2], w3 = wtot[3];
    const u32 nv = w0 + w1 + w2 + w3;
    auto pre_of = [&](u32 chunk) -> u32 {
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
        // LANEROWS: this thread's 16 rows x 16 columns, transposed in registers with v_perm_b32 (two
        // rounds of byte interleaves per 4x4 block, 128 instructions): tr[c][k] = column c, rows 4k..4k+3
        uint32_t tr[(HOLD && LANEROWS) ? 16 : 1][4];
        if constexpr (HOLD && LANEROWS) {
#define EDSX_T(C, COMP)                                                                            \
            _Pragma("unroll") for (int k4 = 0; k4 < 4; k4++) {                                    \
                const uint32_t a0 = d[4 * k4].COMP, a1 = d[4 * k4 + 1].COMP, a2 = d[4 * k4 + 2].COMP, a3 = d[4 * k4 + 3].COMP; \
                const uint32_t t0 = __builtin_amdgcn_perm(a1, a0, 0x05010400u), t1 = __builtin_amdgcn_perm(a1, a0, 0x07030602u); \
                const uint32_t t2 = __builtin_amdgcn_perm(a3, a2, 0x05010400u), t3 = __builtin_amdgcn_perm(a3, a2, 0x07030602u); \
                tr[4 * C][k4] = __builtin_amdgcn_perm(t2, t0, 0x05040100u);                       \
                tr[4 * C + 1][k4] = __builtin_amdgcn_perm(t2, t0, 0x07060302u);                   \
                tr[4 * C + 2][k4] = __builtin_amdgcn_perm(t3, t1, 0x05040100u);                   \
                tr[4 * C + 3][k4] = __builtin_amdgcn_perm(t3, t1, 0x07060302u);                   \
            }
            EDSX_T(0, x) EDSX_T(1, y) EDSX_T(2, z) EDSX_T(3, w)
#undef EDSX_T
        }
        for (u32 b0 = 0; b0 < nv || b0 == 0; b0 += cap) {
            if (V16 && nv) {
                u32 idx = pre_of(j);
                if constexpr (HOLD) {
#define EDSX_X(I)                                                                              \
                    if (V16 & (1u << I)) {                                                     \
                        if (idx >= b0 && idx < b0 + cap) {                                     \
                            uint8_t* dst = colbuf + (size_t)(idx - b0) * p.Spad;               \
                            if constexpr (LANEROWS) { /* RI == 4 * Gp: this thread's 16 rows are 16 */ \
                                constexpr int TI = LANEROWS ? I : 0; /* consecutive bytes of the permuted column */ \
                                *reinterpret_cast<uint4*>(dst + sub * 16) = make_uint4(tr[TI][0], tr[TI][1], tr[TI][2], tr[TI][3]); \
                            } else {                                                           \
                                _Pragma("unroll") for (int it = 0; it < RPT; it++) {           \
                                    const u32 r = sub + it * RI;                               \
                                    if (r < p.S) {                                             \
                                        const u32 ch = byte_at<I>(d[it]);                      \
                                        if (ch == '\n') bad = 1;                               \
                                        dst[vc_pos(r, p.Gp)] = (uint8_t)ch;                    \
                                    }                                                          \
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
                                dst[vc_pos(r, p.Gp)] = (uint8_t)ch;
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

// Cells of one segment.  Read from HBM a cell costs a chain of dependent loads (V word, slot table,
// vc byte); segments of up to STAGE_COLS pure variant columns are first copied into LDS (`st`), which
// turns the generic kernels' latency-bound row walks into LDS reads.
constexpr u32 STAGE_COLS = 64;
struct SegCells {
    const MsaView& mv; u64 a; const uint8_t* st;
    __device__ __forceinline__ u32 at(u64 c, u32 r) const
    {
        if (st) return st[(size_t)(c - a) * mv.Spad + vc_pos(r, mv.Gp)];
        return mv.vbit(c) ? mv.vc[mv.slot(c) * mv.Spad + vc_pos(r, mv.Gp)] : mv.ref_byte(c);
    }
};
// all threads of the workgroup; returns the LDS image of the segment's columns or nullptr
__device__ const uint8_t* stage_columns(const MsaView& mv, u64 a, u64 b, uint8_t* buf, u32 cap_cols, u32* flag_sh)
{
    const u64 ncol = b - a;
    if (!buf || ncol > cap_cols) return nullptr;
    u64* slot_tab = reinterpret_cast<u64*>(buf);              // cap_cols entries, then the columns
    uint8_t* cols = buf + (size_t)cap_cols * 8;
    if (threadIdx.x == 0) *flag_sh = 0;
    __syncthreads();
    if (threadIdx.x < ncol) {
        const u64 c = a + threadIdx.x;
        if (mv.vbit(c)) slot_tab[threadIdx.x] = mv.slot(c);
        else *flag_sh = 1;                                    // a common column inside (context merge): not staged
    }
    __syncthreads();
    if (*flag_sh) return nullptr;
    const u32 vec = mv.Spad / 16;                             // Spad % 16 == 0
    for (u32 i = threadIdx.x; i < (u32)ncol * vec; i += blockDim.x) {
        const u32 c = i / vec, o = (i - c * vec) * 16;
        *reinterpret_cast<uint4*>(cols + (size_t)c * mv.Spad + o) =
            *reinterpret_cast<const uint4*>(mv.vc + slot_tab[c] * (u64)mv.Spad + o);
    }
    __syncthreads();
    return cols;
}

// gap-stripped string of row r over [a,b): the reference drops '\n' and '-' and stops at '\0'
// (msa_transforms.cpp:281-286)
__device__ u64 seg_row_key(const SegCells& sc, u64 a, u64 b, u32 r, bool exact, u32& saw_nl)
{
    u64 key = exact ? 0ull : 0xcbf29ce484222325ull;
    u32 len = 0;
    for (u64 c = a; c < b; c++) {
        u32 ch = sc.at(c, r);
        if (ch == 0) break;
        if (ch == '\n') saw_nl = 1;
        if (ch == '-' || ch == '\n') continue;
        if (exact) key |= (u64)ch << (8 * len);
        else key = (key ^ ch) * 0x100000001b3ull;
        len++;
    }
    if (!exact) key = (key ^ len) * 0x100000001b3ull;
    return key;
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
constexpr u64 HT_EMPTY = ~0ull;
struct HtLds {
    u64* tabk; u32* tabm; u32* bm; u32* pre; u32* flag;
    __device__ HtLds(uint8_t* base)
    {
        tabk = reinterpret_cast<u64*>(base);
        tabm = reinterpret_cast<u32*>(base + (size_t)8 * HT_SIZE);
        bm = reinterpret_cast<u32*>(base + (size_t)12 * HT_SIZE);
        pre = bm + 72;
        flag = pre + 72;
    }
    static constexpr size_t BYTES = (size_t)12 * HT_SIZE + 4 * (72 + 72 + 8);
};

// returns k (number of distinct strings); fills lds.gid[], lds.rep_row[0..k)
__device__ u32 group_segment(const MsaView& mv, u64 a, u64 b, SegLds& lds, u32* rep_sh, const uint8_t* st)
{
    const u32 S = mv.S;
    const SegCells sc{mv, a, st};
    const bool exact = (b - a) <= 8;
    u32 saw_nl = 0;
    for (u32 r = threadIdx.x; r < S; r += GT) {
        lds.key[r] = seg_row_key(sc, a, b, r, exact, saw_nl);
        lds.gid[r] = GID_NONE;
    }
    if (saw_nl) atomicOr(&mv.hdr->status, (u64)(ST_LAYOUT | ST_NEWLINE_IN_DATA));   // a row is ragged

    if (S <= HT_MAX_ROWS) {
        HtLds ht(reinterpret_cast<uint8_t*>(lds.gid + ((S + 7) & ~7u)));
        for (u32 i = threadIdx.x; i < HT_SIZE; i += GT) { ht.tabk[i] = HT_EMPTY; ht.tabm[i] = 0xffffffffu; }
        for (u32 i = threadIdx.x; i < 72; i += GT) ht.bm[i] = 0;
        if (threadIdx.x == 0) *ht.flag = 0;
        __syncthreads();
        for (u32 r = threadIdx.x; r < S; r += GT) {
            u64 kk = lds.key[r];
            if (kk == HT_EMPTY) kk = HT_EMPTY - 1;
            u32 slot = (u32)(mix64(kk) >> 20) & (HT_SIZE - 1);
            while (true) {
                const u64 cur = atomicCAS(&ht.tabk[slot], HT_EMPTY, kk);
                if (cur == HT_EMPTY || cur == kk) { atomicMin(&ht.tabm[slot], r); lds.run[r] = slot; break; }
                slot = (slot + 1) & (HT_SIZE - 1);
            }
        }
        __syncthreads();
        for (u32 r = threadIdx.x; r < S; r += GT) {
            const u32 f = ht.tabm[lds.run[r]];
            if (!exact && f != r && !seg_rows_equal(sc, a, b, r, f)) *ht.flag = 1;
            lds.rep_row[r] = f;                       // temporarily: first row of r's group
            if (f == r) atomicOr(&ht.bm[r >> 5], 1u << (r & 31));
        }
        __syncthreads();
        if (*ht.flag == 0) {
            const u32 nwords = (S + 31) >> 5;
            if (threadIdx.x < nwords) {
                u32 acc = 0;
                for (u32 w = 0; w < threadIdx.x; w++) acc += __builtin_popcount(ht.bm[w]);
                ht.pre[threadIdx.x] = acc;
            }
            if (threadIdx.x == 0) {
                u32 acc = 0;
                for (u32 w = 0; w < nwords; w++) acc += __builtin_popcount(ht.bm[w]);
                *rep_sh = acc;
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
                lds.gid[r] = (uint16_t)myg[j];
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
    const u64* list; const u64* list_n;       // when set: only these segments (left over by the fast path)
    u32 stage_cols = 0, stage_off = 0;        // LDS column staging: capacity in columns, byte offset in the dynamic LDS
};

// K3: per-segment output sizes.  common: "{" ref "}" and "{0}"; variant: see generate_output.
__global__ void __launch_bounds__(GT) k_seg_count(SegParams p)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    __shared__ u32 rep_sh;
    __shared__ u64 sum_sh;
    SegLds lds(lds_raw, p.mv.S);
    if (p.mv.hdr->status) return;                     // vc overflow: the host grows vc and replans
    const u64 nitems = p.list ? *p.list_n : *p.nseg_ptr;
    for (u64 it = blockIdx.x; it < nitems; it += gridDim.x) {
        const u64 seg = p.list ? p.list[it] : it;
        const u64 a = p.seg_start[seg], b = p.seg_start[seg + 1];
        if (!p.mv.vbit(a)) {
            if (threadIdx.x == 0) { p.eds_len[seg] = 2 + (b - a); p.seds_len[seg] = 3; }
            continue;
        }
        if (threadIdx.x == 0) sum_sh = 0;
        const uint8_t* st = stage_columns(p.mv, a, b, p.stage_cols ? lds_raw + p.stage_off : nullptr, p.stage_cols, &rep_sh);
        const u32 k = group_segment(p.mv, a, b, lds, &rep_sh, st);
        const SegCells sc{p.mv, a, st};
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
// K5a: common-segment text.  One wave per 64-column word, lane = column.
//   msa_transforms.cpp:245-258
// ---------------------------------------------------------------------------------------------
struct EmitParams {
    MsaView mv; const u64* seg_start; const u64* nseg_ptr; const u64* Hseg; const u64* segbase;
    const u64* eds_off; const u64* seds_off; uint8_t* eds; uint8_t* seds; u64 nwords;
    const u64* list; const u64* list_n;
    u32 stage_cols = 0, stage_off = 0;
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
    const u64 nitems = p.list ? *p.list_n : *p.nseg_ptr;
    const u32 lane = threadIdx.x & 63;
    for (u64 it = blockIdx.x; it < nitems; it += gridDim.x) {
        const u64 seg = p.list ? p.list[it] : it;
        const u64 a = p.seg_start[seg], b = p.seg_start[seg + 1];
        if (!mv.vbit(a)) continue;
        const uint8_t* st = stage_columns(mv, a, b, p.stage_cols ? lds_raw + p.stage_off : nullptr, p.stage_cols, &rep_sh);
        const u32 k = group_segment(mv, a, b, lds, &rep_sh, st);
        const SegCells sc{mv, a, st};
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

        // ---- seds: wave 0 walks the rows in order; run[g] = next write offset of group g
        if (threadIdx.x < 64) {
            for (u32 base = 0; base < S; base += 64) {
                const u32 r = base + lane;
                const bool valid = r < S;
                const u32 g = valid ? lds.gid[r] : 0xffffffffu;
                const u32 tl = ndigits(r + 1) + 1;
                const u32 tlA = __shfl(tl, 0, 64);              // <= 2 token lengths per 64 rows
                const u64 maskA = ballot64(valid && tl == tlA);
                u64 tok = (u64)',' << (8 * (tl - 1));           // "ddd," little-endian: first digit in byte 0
                { u32 v = r + 1; for (int i = (int)tl - 2; i >= 0; i--) { tok |= (u64)('0' + v % 10u) << (8 * i); v /= 10u; } }
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
                if (valid) {
                    uint8_t* dst = seds + myoff;
                    dst[0] = (uint8_t)tok; dst[1] = (uint8_t)(tok >> 8);
                    if (tl >= 3) dst[2] = (uint8_t)(tok >> 16);
                    if (tl >= 4) dst[3] = (uint8_t)(tok >> 24);
                    if (tl >= 5) dst[4] = (uint8_t)(tok >> 32);
                    if (tl >= 6) for (u32 i = 5; i < tl; i++) dst[i] = (uint8_t)(tok >> (8 * i));
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
// Fast path (S <= 1024): one WAVE per variant segment, rows in registers.
//   lane holds rows lane, lane+64, ... (byte i of its 16-byte vc load = row i*64+lane), so row
//   order = (i, lane) and consecutive ids sit in consecutive lanes.  Per-row state is SWAR bytes
//   in a uint4: gid (group id), rm (rows not yet grouped).
//   Grouping: rows are first grouped by RAW equality over the segment's columns ('-' and '\n'
//   normalised to 0), which needs no per-row gap stripping; the gap-stripped string of each group
//   is then built once, in scalar registers, from its representative.  Two raw groups with the
//   same stripped string (same letters, different gap placement), more than KMAX groups or more
//   than 16 columns send the segment to the generic workgroup-per-segment kernels (slow_list).
// ---------------------------------------------------------------------------------------------
constexpr int KMAX = 8;
constexpr u64 META_FAST = 1ull << 63;     // | ncol << 48 | slot of the first column
constexpr u64 META_SCATTER = 1ull << 61;  // slots of the columns are not consecutive (tile edge)
constexpr u64 META_SLOT = 0xffffffffffffull;

// thread per segment: sizes of common segments, slot/ncol descriptor of variant segments
__global__ void __launch_bounds__(256) k_seg_meta(FastParams p)
{
    const MsaView& mv = p.mv;
    const u64 nseg = *p.nseg_ptr;
    for (u64 seg = blockIdx.x * (u64)blockDim.x + threadIdx.x; seg < nseg; seg += (u64)gridDim.x * blockDim.x) {
        const u64 a = p.seg_start[seg], b = p.seg_start[seg + 1];
        if (!mv.vbit(a)) {
            p.eds_len[seg] = 2 + (b - a);
            p.seds_len[seg] = 3;
            p.segmeta[seg] = 0;
            continue;
        }
        u64 meta = 0;                                     // 0 on a variant segment: not fast
        const u64 ncol = b - a;
        if (ncol <= 64) {
            const u64 s0 = mv.slot(a);
            bool pure = true, contig = true;
            for (u64 c = a + 1; c < b; c++) {
                pure = pure && mv.vbit(c);
                if (pure) contig = contig && mv.slot(c) == s0 + (c - a);
            }
            if (pure) meta = META_FAST | (contig ? 0 : META_SCATTER) | (ncol << 48) | s0;
        }
        if (!meta) p.slow_list[atomicAdd(p.slow_count, 1ull)] = seg;      // too wide for the fast kernels
        p.segmeta[seg] = meta;
    }
}

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

constexpr int KCAP = 64;          // distinct strings per fast segment (group g lives in lane g)
// Grouping record of a fast segment, written by the count kernel and read by the emit kernel
// (so the rows are grouped once): 1024 group-id bytes in vc row order | rep[64] u16 | letter[64] u8
// | k.  Indexed by the ordinal of the variant segment.
constexpr u32 GREC_BYTES = 1280, GREC_REP = 1024, GREC_CHR = 1152, GREC_K = 1216;
struct FastGroups {
    uint4 gid;            // byte i = group of row i*64+lane (0xFF: no such row)
    u32 k;                // number of distinct strings (wave-uniform)
    u32 sumlen;           // sum of their lengths
    // lane g holds the state of group g
    u64 key_lo, key_hi;   // ncol == 1: the letter; else 96-bit hash of the gap-stripped string
    u32 rep;              // representative row (first row of the group in row order)
    u32 len;              // length of the group's string
};

// lane-private validity mask: byte i = 0xFF iff row i*64+lane exists
__device__ __forceinline__ uint4 fast_valid_mask(u32 lane, u32 S)
{
    uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if ((u32)i * 64u + lane < S) w0 |= 0xffu << (i * 8);
        if ((u32)(i + 4) * 64u + lane < S) w1 |= 0xffu << (i * 8);
        if ((u32)(i + 8) * 64u + lane < S) w2 |= 0xffu << (i * 8);
        if ((u32)(i + 12) * 64u + lane < S) w3 |= 0xffu << (i * 8);
    }
    return make_uint4(w0, w1, w2, w3);
}

// the not yet grouped row that comes first in row order (i, lane): returns i (uniform) and its lane
__device__ __forceinline__ u32 first_remaining(const uint4& rm, int& leader)
{
    const u32 i0 = first_byte_index(rm);
    u64 bl = 0;
    u32 t = 0;
    for (; t < 16; t++) { bl = ballot64(i0 == t); if (bl) break; }
    leader = __builtin_ctzll(bl);
    return t;
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

// XOR of v over the 64 lanes, wave-uniform (DPP row shifts and broadcasts; lane 63 ends up with all)
template <int CTRL, int ROWMASK> __device__ __forceinline__ u32 dpp_move0(u32 v)
{
    return (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWMASK, 0xf, false);   // lanes without a source get 0
}
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
// minimum of v over lanes 0..7 (wave-uniform)
__device__ __forceinline__ u32 min_lanes8(u32 v)
{
    u32 t;
    t = (u32)__builtin_amdgcn_update_dpp(-1, (int)v, 0x111, 0xf, 0xf, false); v = t < v ? t : v;
    t = (u32)__builtin_amdgcn_update_dpp(-1, (int)v, 0x112, 0xf, 0xf, false); v = t < v ? t : v;
    t = (u32)__builtin_amdgcn_update_dpp(-1, (int)v, 0x114, 0xf, 0xf, false); v = t < v ? t : v;
    return (u32)__builtin_amdgcn_readlane((int)v, 7);
}

// One column over the alphabet {A, C, G, T, N, -}: the grouping of msa_transforms.cpp:262-293 with
// byte-table lookups (v_perm_b32 = four lookups in an 8-entry table per instruction).
//   class(b) = ((b >> 1) ^ (b >> 2)) & 7 :  A 0, C 1, G 2, N 4, '-' 5, T 7  (3 and 6 unused)
// Any other byte (lower case, IUPAC codes, a stray newline) fails the reverse lookup and the caller
// takes the general path.  Group ids are the ranks of the classes by first row; the groups' state
// (representative row, letter) lands in lanes 0..k-1 as in fast_assign.
__device__ __forceinline__ bool fast_group_dna1(const uint4& x, const uint4& vmask, u32 lane, FastGroups& G)
{
    constexpr u32 LET_LO = 0x00474341u, LET_HI = 0x54002d4eu;      // class -> letter (0: unused class)
    constexpr u32 OH_LO = 0x08040201u, OH_HI = 0x80402010u;        // class -> 1 << class
    uint4 cls;
    cls.x = ((x.x >> 1) ^ (x.x >> 2)) & 0x07070707u; cls.y = ((x.y >> 1) ^ (x.y >> 2)) & 0x07070707u;
    cls.z = ((x.z >> 1) ^ (x.z >> 2)) & 0x07070707u; cls.w = ((x.w >> 1) ^ (x.w >> 2)) & 0x07070707u;
    const u32 bad = ((__builtin_amdgcn_perm(LET_HI, LET_LO, cls.x) ^ x.x) & vmask.x) |
                    ((__builtin_amdgcn_perm(LET_HI, LET_LO, cls.y) ^ x.y) & vmask.y) |
                    ((__builtin_amdgcn_perm(LET_HI, LET_LO, cls.z) ^ x.z) & vmask.z) |
                    ((__builtin_amdgcn_perm(LET_HI, LET_LO, cls.w) ^ x.w) & vmask.w);
    if (ballot64(bad != 0)) return false;
    // classes present in the column
    u32 pl = (__builtin_amdgcn_perm(OH_HI, OH_LO, cls.x) & vmask.x) | (__builtin_amdgcn_perm(OH_HI, OH_LO, cls.y) & vmask.y) |
             (__builtin_amdgcn_perm(OH_HI, OH_LO, cls.z) & vmask.z) | (__builtin_amdgcn_perm(OH_HI, OH_LO, cls.w) & vmask.w);
    pl |= pl >> 16; pl |= pl >> 8;
    const u32 P = wave_or_all(pl & 0xffu);
    cls.x |= ~vmask.x; cls.y |= ~vmask.y; cls.z |= ~vmask.z; cls.w |= ~vmask.w;   // rows that do not exist: 0xFF
    // first row of every class: lane c keeps class c's (rows are scanned 64 at a time, in row order)
    u32 firstv = 0xffffffffu, missing = P;
#define EDSX_F(I)                                                                                 \
    if (missing) {                                                                                \
        const u32 cb = byte_at<I>(cls);                                                           \
        for (u32 mm = missing; mm; mm &= mm - 1) {                                                \
            const u32 c = (u32)__builtin_ctz(mm);                                                 \
            const u64 b = ballot64(cb == c);                                                      \
            if (b) { firstv = lane == c ? (u32)I * 64u + (u32)__builtin_ctzll(b) : firstv; missing &= ~(1u << c); } \
        }                                                                                         \
    }
    EDSX_F(0) EDSX_F(1) EDSX_F(2) EDSX_F(3) EDSX_F(4) EDSX_F(5) EDSX_F(6) EDSX_F(7)
    EDSX_F(8) EDSX_F(9) EDSX_F(10) EDSX_F(11) EDSX_F(12) EDSX_F(13) EDSX_F(14) EDSX_F(15)
#undef EDSX_F
    // classes in order of first row -> group ids; table class -> group for the lookup below
    u64 lut = ~0ull;
    u32 g = 0, sumlen = 0;
    G.key_lo = 0; G.key_hi = 0; G.rep = 0; G.len = 0;
    for (u32 rem = P; rem; g++) {
        const u32 mn = min_lanes8(firstv);
        const u32 c = (u32)__builtin_ctzll(ballot64(lane < 8u && firstv == mn));
        const u32 letter = c == 5u ? 0u : (u32)((((u64)LET_HI << 32) | LET_LO) >> (8u * c)) & 0xffu;
        lut = (lut & ~(0xffull << (8u * c))) | ((u64)g << (8u * c));
        if (lane == g) { G.key_lo = letter; G.rep = mn; G.len = letter ? 1u : 0u; }
        sumlen += letter ? 1u : 0u;
        firstv = lane == c ? 0xffffffffu : firstv;
        rem &= ~(1u << c);
    }
    G.k = g; G.sumlen = sumlen;
    const u32 lut_lo = (u32)lut, lut_hi = (u32)(lut >> 32);
    G.gid = make_uint4(__builtin_amdgcn_perm(lut_hi, lut_lo, cls.x), __builtin_amdgcn_perm(lut_hi, lut_lo, cls.y),
                       __builtin_amdgcn_perm(lut_hi, lut_lo, cls.z), __builtin_amdgcn_perm(lut_hi, lut_lo, cls.w));
    return true;
}

// Group the rows of a fast segment (msa_transforms.cpp:262-293: distinct gap-stripped strings in
// order of first appearance).  Returns false when the segment must take the generic path.
//   one column : exact, SWAR byte compares.
//   2..10 cols over {A,C,G,T,N,-}: exact 3-bit-per-column keys.
//   otherwise  : every row gets a 96-bit additive signature of its raw column bytes (gaps normalised
//                to 0; v_mad_u32_u24 = full rate); rows with equal signatures are PROPOSED as a raw
//                group and then compared with the group's first row byte for byte (phase B), so the
//                grouping is exact; raw groups that spell the same string are joined by their
//                stripped string (verbatim key up to 12 letters; longer strings that hash alike send
//                the segment to the generic kernels).  NUL bytes (msa_transforms.cpp:282) -> generic.
// NR: rows per lane that can exist (16; 4 when S <= 256)
template <bool CHECK_NL, int NR>
__device__ __forceinline__ bool fast_group(const MsaView& mv, u64 seg_a, u64 meta, const uint4& col0, u32 lane,
                                           const uint4& vmask, FastGroups& G, u32& saw_nl)
{
    const u32 ncol = (u32)(meta >> 48) & 0xffu;
    uint4 rm = vmask;
    G.gid = make_uint4(~0u, ~0u, ~0u, ~0u);
    G.k = 0; G.sumlen = 0; G.key_lo = 0; G.key_hi = 0; G.rep = 0; G.len = 0;

    if (ncol == 1) {
        if (fast_group_dna1(col0, vmask, lane, G)) return true;
        G.gid = make_uint4(~0u, ~0u, ~0u, ~0u);
        G.k = 0; G.sumlen = 0; G.key_lo = 0; G.key_hi = 0; G.rep = 0; G.len = 0;
        // a NUL byte ends the row's string in the reference (msa_transforms.cpp:282): exact kernels only
        if (ballot64(any_nul(col0, vmask))) return false;
        const uint4 col = normalise_col<CHECK_NL>(col0, vmask, saw_nl);
        while (ballot64(any4(rm))) {
            int leader;
            const u32 i0 = first_remaining(rm, leader);
            const u32 c = leader_byte(col, leader, i0);
            uint4 eq = bytes_eq_mask(col, c * 0x01010101u);
            eq.x &= rm.x; eq.y &= rm.y; eq.z &= rm.z; eq.w &= rm.w;
            if (!fast_assign(G, rm, eq, (u64)c, 0ull, c ? 1u : 0u, lane, i0 * 64u + (u32)leader)) return false;
        }
        return true;
    }

    const u64 slot0 = meta & META_SLOT;
    const bool scatter = (meta & META_SCATTER) != 0;
    const uint8_t* cbase = mv.vc + slot0 * (u64)mv.Spad + (u64)lane * mv.Gp;
    auto col_ptr = [&](u32 c) -> const uint8_t* {
        return scatter ? mv.vc + mv.slot(seg_a + c) * (u64)mv.Spad + (u64)lane * mv.Gp : cbase + (u64)c * mv.Spad;
    };

    // ---- up to 10 columns over {A,C,G,T,N,-}: EXACT raw keys, 3 bits per column (class code of
    // fast_group_dna1), one dword per row.  Rows with equal keys are byte-identical; a raw group's
    // gap-stripped string is read off its key (drop the gap classes), so no row is re-read.
    if (ncol <= 10u) {
        constexpr u32 LET_LO = 0x00474341u, LET_HI = 0x54002d4eu;
        u32 key[16];
#pragma unroll
        for (int i = 0; i < 16; i++) key[i] = 0;
        u32 badacc = 0;
#define EDSX_K(I) if constexpr (I < NR) key[I] |= byte_at<I>(cls) << sh;
        for (u32 c0 = 0; c0 < ncol; c0 += 4) {
            uint4 cvs[4];                              // four column loads in flight
#pragma unroll
            for (int j = 0; j < 4; j++) {
                cvs[j] = make_uint4(0, 0, 0, 0);
                if (c0 + j < ncol) cvs[j] = (c0 + j == 0) ? col0 : load16u(col_ptr(c0 + j));
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (c0 + j < ncol) {
                    const uint4 x = cvs[j];
                    uint4 cls;
                    cls.x = ((x.x >> 1) ^ (x.x >> 2)) & 0x07070707u; cls.y = ((x.y >> 1) ^ (x.y >> 2)) & 0x07070707u;
                    cls.z = ((x.z >> 1) ^ (x.z >> 2)) & 0x07070707u; cls.w = ((x.w >> 1) ^ (x.w >> 2)) & 0x07070707u;
                    badacc |= ((__builtin_amdgcn_perm(LET_HI, LET_LO, cls.x) ^ x.x) & vmask.x) |
                              ((__builtin_amdgcn_perm(LET_HI, LET_LO, cls.y) ^ x.y) & vmask.y) |
                              ((__builtin_amdgcn_perm(LET_HI, LET_LO, cls.z) ^ x.z) & vmask.z) |
                              ((__builtin_amdgcn_perm(LET_HI, LET_LO, cls.w) ^ x.w) & vmask.w);
                    const u32 sh = 3u * (c0 + j);
                    EDSX_K(0) EDSX_K(1) EDSX_K(2) EDSX_K(3) EDSX_K(4) EDSX_K(5) EDSX_K(6) EDSX_K(7)
                    EDSX_K(8) EDSX_K(9) EDSX_K(10) EDSX_K(11) EDSX_K(12) EDSX_K(13) EDSX_K(14) EDSX_K(15)
                }
            }
        }
#undef EDSX_K
        if (!ballot64(badacc != 0)) {
            while (ballot64(any4(rm))) {
                int leader;
                const u32 i0 = first_remaining(rm, leader);
                u32 mk = 0;
#pragma unroll
                for (int i = 0; i < NR; i++) mk = (i0 == (u32)i) ? key[i] : mk;
                const u32 rk = (u32)__builtin_amdgcn_readlane((int)mk, leader);
                uint32_t e0 = 0, e1 = 0, e2 = 0, e3 = 0;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    if (key[i] == rk) e0 |= 0xffu << (i * 8);
                    if constexpr (NR > 4) {
                        if (key[i + 4] == rk) e1 |= 0xffu << (i * 8);
                        if (key[i + 8] == rk) e2 |= 0xffu << (i * 8);
                        if (key[i + 12] == rk) e3 |= 0xffu << (i * 8);
                    }
                }
                const uint4 eq = make_uint4(e0 & rm.x, e1 & rm.y, e2 & rm.z, e3 & rm.w);
                // the group's string: its key without the gap classes (wave-uniform scalar work)
                u64 sk = 0;
                u32 len = 0;
                for (u32 c = 0; c < ncol; c++) {
                    const u32 cl = (rk >> (3u * c)) & 7u;
                    if (cl != 5u) { sk |= (u64)cl << (3u * len); len++; }
                }
                if (!fast_assign(G, rm, eq, sk, (u64)len << 32, len, lane, i0 * 64u + (u32)leader)) return false;
            }
            return true;
        }
        rm = vmask;                                    // another alphabet: the signature path below
    }

    // ---- phase A: raw signatures of the 16 rows of this lane: sig_j(row) = sum_c byte(row,c) * W_j(c)
    // (gaps are 0 and contribute nothing; v_mad_u32_u24 is full rate).  Rows with equal normalised
    // columns get equal signatures; a collision of different rows is caught by the comparison in phase B.
    u32 h1[16], h2[16], h3[16];
    u32 nul = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) { h1[i] = 0; h2[i] = 0; h3[i] = 0; }
    auto weight = [](u32 c, u32 j) -> u32 { return FAST_W.v[c * 3u + j]; };
#define EDSX_H(I)                                                                                 \
        {                                                                                         \
            const u32 bch = byte_at<I>(cn);                                                       \
            h1[I] += __umul24(bch, w1); h2[I] += __umul24(bch, w2); h3[I] += __umul24(bch, w3);   \
        }
    for (u32 c0 = 0; c0 < ncol; c0 += 4) {
        uint4 cvs[4];                                  // four column loads in flight
#pragma unroll
        for (int j = 0; j < 4; j++) {
            cvs[j] = make_uint4(0, 0, 0, 0);
            if (c0 + j < ncol) cvs[j] = (c0 + j == 0) ? col0 : load16u(col_ptr(c0 + j));
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (c0 + j < ncol) {
                nul |= any_nul(cvs[j], vmask) ? 1u : 0u;
                const uint4 cn = normalise_col<CHECK_NL>(cvs[j], vmask, saw_nl);
                const u32 w1 = weight(c0 + j, 0), w2 = weight(c0 + j, 1), w3 = weight(c0 + j, 2);
                EDSX_H(0) EDSX_H(1) EDSX_H(2) EDSX_H(3) EDSX_H(4) EDSX_H(5) EDSX_H(6) EDSX_H(7)
                EDSX_H(8) EDSX_H(9) EDSX_H(10) EDSX_H(11) EDSX_H(12) EDSX_H(13) EDSX_H(14) EDSX_H(15)
            }
        }
    }
#undef EDSX_H
    if (ballot64(nul != 0)) return false;              // msa_transforms.cpp:282: left to the exact generic kernels
#ifdef EDSX_TEST_WEAK_SIG
    // test-only build: one bit of signature, so that different rows collide all the time and the
    // byte-for-byte verification in phase B is what keeps the groups right
#pragma unroll
    for (int i = 0; i < 16; i++) { h1[i] &= 1u; h2[i] = 0; h3[i] = 0; }
#endif
    // ---- phase B: raw groups in order of first appearance; each is keyed by a 96-bit hash of its
    // representative's gap-stripped string (+ length), so that raw groups spelling the same string
    // (same letters, other gap placement) fall together in fast_assign
    while (ballot64(any4(rm))) {
        int leader;
        const u32 i0 = first_remaining(rm, leader);
        u32 m1 = 0, m2 = 0, m3 = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            m1 = (i0 == (u32)i) ? h1[i] : m1; m2 = (i0 == (u32)i) ? h2[i] : m2; m3 = (i0 == (u32)i) ? h3[i] : m3;
        }
        const u32 r1 = (u32)__builtin_amdgcn_readlane((int)m1, leader), r2 = (u32)__builtin_amdgcn_readlane((int)m2, leader);
        const u32 r3 = (u32)__builtin_amdgcn_readlane((int)m3, leader);
        uint32_t e0 = 0, e1 = 0, e2 = 0, e3 = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (h1[i] == r1 && h2[i] == r2 && h3[i] == r3) e0 |= 0xffu << (i * 8);
            if (h1[i + 4] == r1 && h2[i + 4] == r2 && h3[i + 4] == r3) e1 |= 0xffu << (i * 8);
            if (h1[i + 8] == r1 && h2[i + 8] == r2 && h3[i + 8] == r3) e2 |= 0xffu << (i * 8);
            if (h1[i + 12] == r1 && h2[i + 12] == r2 && h3[i + 12] == r3) e3 |= 0xffu << (i * 8);
        }
        const uint4 eq = make_uint4(e0 & rm.x, e1 & rm.y, e2 & rm.z, e3 & rm.w);
        // The signature only proposes the group: every member is now compared with the leader byte for
        // byte over all columns (gaps normalised as above).  A mismatch is a signature collision: the
        // segment goes to the generic kernels, which compare rows exactly.
        {
            u32 mism = 0;
            for (u32 c = 0; c < ncol; c++) {               // (one load at a time: this path is rare, registers are not)
                u32 dummy = 0;
                const uint4 cn = normalise_col<false>(c == 0 ? col0 : load16u(col_ptr(c)), vmask, dummy);
                const u32 lb = leader_byte(cn, leader, i0) * 0x01010101u;
                mism |= ((cn.x ^ lb) & eq.x) | ((cn.y ^ lb) & eq.y) | ((cn.z ^ lb) & eq.z) | ((cn.w ^ lb) & eq.w);
            }
            if (ballot64(mism != 0)) return false;
        }
        // the representative's string: lane = column
        const u32 rep_row = i0 * 64u + (u32)leader;
        u32 ch = 0;
        if (lane < ncol) {
            const u64 sl = scatter ? mv.slot(seg_a + lane) : slot0 + lane;
            ch = mv.vc[sl * (u64)mv.Spad + vc_pos(rep_row, mv.Gp)];
            if (ch == '-' || ch == '\n') ch = 0;
        }
        const u64 nzm = ballot64(ch != 0);
        const u32 len = (u32)__builtin_popcountll(nzm);
        const u32 pos = mbcnt(nzm);
        // key: the first 12 letters verbatim (96 bits: exact for strings up to 12 letters), the
        // letters behind them hashed on top; the length goes into the key separately
        u32 t1 = 0, t2 = 0, t3 = 0;
        if (ch && pos < 12u) {
            const u32 v = ch << ((pos & 3u) * 8u), wsel = pos >> 2;
            t1 = wsel == 0u ? v : 0u; t2 = wsel == 1u ? v : 0u; t3 = wsel == 2u ? v : 0u;
        }
        if (len > 12u) {                                   // wave-uniform
            if (ch && pos >= 12u) {
                u32 x = (ch + 1u) * 0x9e3779b1u ^ (pos + 1u) * 0x85ebca77u;
                x ^= x >> 16; x *= 0x21f0aaadu; x ^= x >> 15;
                t1 = x * 0x735a2d97u; t1 ^= t1 >> 15;
                t2 = (x ^ 0x5bd1e995u) * 0xc2b2ae3du; t2 ^= t2 >> 13;
                t3 = (x + 0x27d4eb2fu) * 0x165667b1u; t3 ^= t3 >> 16;
            }
        }
        t1 = wave_xor_all(t1); t2 = wave_xor_all(t2); t3 = wave_xor_all(t3);
#ifdef EDSX_TEST_WEAK_SIG
        if (len > 12u) { t1 &= 1u; t2 = 0; t3 = 0; }
#endif
        const u64 klo = ((u64)t2 << 32) | t1, khi = ((u64)len << 32) | t3;
        // Keys of up to 12 letters are the string itself.  Longer ones are hashed: an equal key then only
        // SUGGESTS that two raw groups (same letters, other gap placement) spell one string, so such a
        // segment is left to the exact generic kernels.
        if (len > 12u && ballot64(lane < G.k && G.key_lo == klo && G.key_hi == khi)) return false;
        if (!fast_assign(G, rm, eq, klo, khi, len, lane, rep_row)) return false;
    }
    return true;
}

__device__ __forceinline__ u32 packed_len(u64 k) { return k ? (u32)(71 - __builtin_clzll(k)) >> 3 : 0u; }

__device__ __forceinline__ uint4 fast_load_col(const MsaView& mv, u64 meta, u32 lane)
{
    if (!(meta & META_FAST)) return make_uint4(0, 0, 0, 0);
    return load16u(mv.vc + (meta & META_SLOT) * (u64)mv.Spad + (u64)lane * mv.Gp);
}


// K3 fast: sizes of the variant segments.  Variant and common segments alternate, so the
// variant ones are seg = 2*vi + p0.
template <int NR>
__global__ void __launch_bounds__(256, 4) k_seg_count_fast(FastParams p)
{
    const MsaView& mv = p.mv;
    if (mv.hdr->status) return;
    const u32 lane = threadIdx.x & 63;
    const u64 nseg = *p.nseg_ptr;
    const u64 p0 = mv.vbit(0) ? 0 : 1;
    const u64 nvs = nseg > p0 ? (nseg - p0 + 1) / 2 : 0;
    const u64 nw = ((u64)gridDim.x * blockDim.x) >> 6;
    const uint4 vmask = fast_valid_mask(lane, mv.S);
    u32 saw_nl = 0;
    // everything indexed by the segment is wave-uniform: say so (readfirstlane), or hipcc predicates
    // every use per lane and serialises the column loads
    u64 vi = (u64)blockIdx.x * (blockDim.x >> 6) + uniform32(threadIdx.x >> 6);
    u64 meta = vi < nvs ? uniform64(p.segmeta[2 * vi + p0]) : 0;
    u64 meta_n = vi + nw < nvs ? uniform64(p.segmeta[2 * (vi + nw) + p0]) : 0;
    uint4 col = fast_load_col(mv, meta, lane);
    while (vi < nvs) {
        const u64 seg = 2 * vi + p0;
        // prefetch: next segment's first column and the descriptor after it
        const uint4 col_n = fast_load_col(mv, meta_n, lane);
        const u64 meta_v = p.segmeta[2 * (vi + 2 * nw < nvs ? vi + 2 * nw : vi) + p0];   // scalar after the wait below

        bool fast = (meta & META_FAST) != 0;
        FastGroups G;
        if (fast) fast = fast_group<true, NR>(mv, (meta & META_SCATTER) ? uniform64(p.seg_start[seg]) : 0, meta, col, lane, vmask, G, saw_nl);
        uint8_t* rec = p.grec + vi * (u64)GREC_BYTES;
        // wait for the prefetched column here, before this segment's stores are queued behind it
        // (vmcnt retires in issue order: see k_emit_variant_fast)
        asm volatile("" :: "v"(col_n.x), "v"(col_n.y), "v"(col_n.z), "v"(col_n.w), "v"(meta_v));
        const u64 meta_nn = vi + 2 * nw < nvs ? uniform64(meta_v) : 0;
        if (fast) {
            *reinterpret_cast<uint4*>(rec + lane * 16u) = G.gid;
            if (lane < G.k) {
                *reinterpret_cast<uint16_t*>(rec + GREC_REP + lane * 2u) = (uint16_t)G.rep;
                rec[GREC_CHR + lane] = (uint8_t)G.key_lo;
            }
            if (lane == 0) {
                rec[GREC_K] = (uint8_t)G.k;
                p.eds_len[seg] = 2 + (u64)(G.k - 1) + G.sumlen;
                p.seds_len[seg] = (u64)G.k + p.tok_total;
            }
        } else if (lane == 0) {
            rec[GREC_K] = 0;                              // not a fast segment
            if (meta & META_FAST) p.slow_list2[atomicAdd(p.slow_count2, 1ull)] = seg;   // gave up: too many strings
        }
        vi += nw; meta = meta_n; meta_n = meta_nn; col = col_n;
    }
    if (saw_nl) atomicOr(&mv.hdr->status, (u64)(ST_LAYOUT | ST_NEWLINE_IN_DATA));
}

// ---- K5 fast: .seds / .eds text of the variant segments.  msa_transforms.cpp:297-317.
//   LDS per workgroup: token table tok[1024] (u64: "ddd," + length in byte 7) | per-wave staging.
//   Ids of block i (rows i*64 .. i*64+63) are i*64+1 .. i*64+64: 3 digits for blocks 2..14; the
//   digit count changes inside block 0 (ids 1-9 | 10-64), block 1 (65-99 | 100-128) and block 15
//   (961-999 | 1000-1024): there the lanes below SHORT have the shorter token.
constexpr int FAST_STAGE = 4608;   // >= 16 + KCAP + tokens of 1024 rows (4013 + 250 + 16), multiple of 256

template <int I> struct IdBlock {
    static constexpr u32 TLMIN = I == 0 ? 2u : (I == 1 ? 3u : 4u);
    static constexpr bool MIXED = I == 0 || I == 1 || I == 15;
    static constexpr u64 SHORT = I == 0 ? ((1ull << 9) - 1) : (I == 1 ? ((1ull << 35) - 1) : (I == 15 ? ((1ull << 39) - 1) : ~0ull));
};

__device__ __forceinline__ u32 gid_byte(const uint4& gid, u32 i)       // i is wave-uniform
{
    const u32 w = i < 8 ? (i < 4 ? gid.x : gid.y) : (i < 12 ? gid.z : gid.w);
    return (w >> ((i & 3) * 8)) & 0xffu;
}

// one block of 64 rows: where does this lane's token go, and advance the group cursors.
//   cursors: registers (K compile-time, <= 4) ...
template <int K, bool MIXED>
__device__ __forceinline__ u32 block_offsets(u32 gi, u32 (&cur)[4], u32 tlmin, u64 shortm)
{
    u32 off = 0;
#pragma unroll
    for (int g = 0; g < K; g++) {
        const u64 m = ballot64(gi == (u32)g);
        u32 pre = mbcnt(m) * tlmin;
        u32 tot = (u32)__builtin_popcountll(m) * tlmin;
        if (MIXED) { pre += mbcnt(m & ~shortm); tot += (u32)__builtin_popcountll(m & ~shortm); }
        off = (gi == (u32)g) ? cur[g] + pre : off;
        cur[g] += tot;
    }
    return off;
}
//   ... or lane g's register cur_l (K run-time)
template <bool MIXED>
__device__ __forceinline__ u32 block_offsets_dyn(u32 gi, u32 k, u32& cur_l, u32 tlmin, u64 shortm, u32 lane)
{
    u32 off = 0;
    for (u32 g = 0; g < k; g++) {
        const u64 m = ballot64(gi == g);
        if (!m) continue;
        const u32 c = (u32)__builtin_amdgcn_readlane((int)cur_l, (int)g);
        u32 pre = mbcnt(m) * tlmin;
        u32 tot = (u32)__builtin_popcountll(m) * tlmin;
        if (MIXED) { pre += mbcnt(m & ~shortm); tot += (u32)__builtin_popcountll(m & ~shortm); }
        off = (gi == g) ? c + pre : off;
        cur_l = (lane == g) ? c + tot : cur_l;
    }
    return off;
}

// Token bytes -> LDS as single-byte stores.  Written in C (d[0] = ..; d[1] = ..) hipcc fuses the four
// stores into one ds_write_b32 at an unaligned address, which the LDS executes ~10x slower
// (SQ_LDS_IDX_ACTIVE: 28 cycles per instruction).  a = LDS byte address.
__device__ __forceinline__ void lds_put2(u32 a, u32 t)
{
    const u32 t8 = t >> 8;
    asm volatile("ds_write_b8 %0, %1\n\tds_write_b8 %0, %2 offset:1" :: "v"(a), "v"(t), "v"(t8) : "memory");
}
__device__ __forceinline__ void lds_put4(u32 a, u32 t)
{
    const u32 t8 = t >> 8;          // d16_hi stores bits 23:16: bytes 2 and 3 need no further shifts
    asm volatile("ds_write_b8 %0, %1\n\tds_write_b8 %0, %2 offset:1\n\t"
                 "ds_write_b8_d16_hi %0, %1 offset:2\n\tds_write_b8_d16_hi %0, %2 offset:3"
                 :: "v"(a), "v"(t), "v"(t8) : "memory");
}
template <int OFF> __device__ __forceinline__ void lds_put1(u32 a, u32 v)
{
    asm volatile("ds_write_b8 %0, %1 offset:%2" :: "v"(a), "v"(v), "n"(OFF) : "memory");
}
template <bool MIXED>
__device__ __forceinline__ void write_token(uint8_t* dst, u64 t)
{
    const u32 a = (u32)(uintptr_t)dst;    // LDS byte address (low half of the flat address)
    if (!MIXED) { lds_put4(a, (u32)t); return; }
    const u32 tl = (u32)(t >> 56);
    lds_put2(a, (u32)t);
    if (tl >= 3) lds_put1<2>(a, (u32)(t >> 16));
    if (tl >= 4) lds_put1<3>(a, (u32)(t >> 24));
    if (tl >= 5) lds_put1<4>(a, (u32)(t >> 32));
}

// Writes the id lists of the k groups into `text`; returns the number of bytes (= k + tokens).
// K > 0: compile-time group count with cursors in registers; K == 0: run-time k (<= 64), the
// cursor of group g lives in lane g.
template <int K>
__device__ __forceinline__ u32 fast_emit_ids(const uint4& gid, u32 k, uint8_t* text, const u64* tok_sh, u32 lane)
{
    u32 cur[4] = {0, 0, 0, 0};
    u32 cur_l = 0, gstart_l = 0;
    // pass 1: bytes per group
    if (K) {
        block_offsets<K, true>(gid_byte(gid, 0), cur, IdBlock<0>::TLMIN, IdBlock<0>::SHORT);
        block_offsets<K, true>(gid_byte(gid, 1), cur, IdBlock<1>::TLMIN, IdBlock<1>::SHORT);
        for (u32 i = 2; i < 15; i++) block_offsets<K, false>(gid_byte(gid, i), cur, 4u, ~0ull);
        block_offsets<K, true>(gid_byte(gid, 15), cur, IdBlock<15>::TLMIN, IdBlock<15>::SHORT);
    } else {
        block_offsets_dyn<true>(gid_byte(gid, 0), k, cur_l, IdBlock<0>::TLMIN, IdBlock<0>::SHORT, lane);
        block_offsets_dyn<true>(gid_byte(gid, 1), k, cur_l, IdBlock<1>::TLMIN, IdBlock<1>::SHORT, lane);
        for (u32 i = 2; i < 15; i++) block_offsets_dyn<false>(gid_byte(gid, i), k, cur_l, 4u, ~0ull, lane);
        block_offsets_dyn<true>(gid_byte(gid, 15), k, cur_l, IdBlock<15>::TLMIN, IdBlock<15>::SHORT, lane);
    }
    // group starts
    u32 run = 0;
    u32 gstart[4] = {0, 0, 0, 0};
    if (K) {
#pragma unroll
        for (int g = 0; g < K; g++) { const u32 t = cur[g]; gstart[g] = run; cur[g] = run + 1; run += 1 + t; }
    } else {
        const u32 mine = lane < k ? 1u + cur_l : 0u;      // bytes of group `lane`
        u32 incl = mine;
        for (int o = 1; o < 64; o <<= 1) { u32 a = __shfl_up(incl, o, 64); if (lane >= (u32)o) incl += a; }
        gstart_l = incl - mine;
        cur_l = gstart_l + 1;
        run = (u32)__builtin_amdgcn_readlane((int)incl, 63);
    }
    // pass 2: tokens, all lanes active, exact-length LDS writes
#define EDSX_BLK(I, MIXED_)                                                                       \
    {                                                                                             \
        const u32 gi = gid_byte(gid, I);                                                          \
        const u32 off = K ? block_offsets<K, MIXED_>(gi, cur, MIXED_ ? IdBlock<(I) & 15>::TLMIN : 4u, \
                                                     MIXED_ ? IdBlock<(I) & 15>::SHORT : ~0ull)   \
                          : block_offsets_dyn<MIXED_>(gi, k, cur_l, MIXED_ ? IdBlock<(I) & 15>::TLMIN : 4u, \
                                                      MIXED_ ? IdBlock<(I) & 15>::SHORT : ~0ull, lane); \
        if (gi != 0xffu) write_token<MIXED_>(text + off, tok_sh[(I) * 64u + lane]);               \
    }
    EDSX_BLK(0, true)
    EDSX_BLK(1, true)
    for (u32 i = 2; i < 15; i++) {
        const u32 gi = gid_byte(gid, i);
        const u32 off = K ? block_offsets<K, false>(gi, cur, 4u, ~0ull) : block_offsets_dyn<false>(gi, k, cur_l, 4u, ~0ull, lane);
        if (gi != 0xffu) write_token<false>(text + off, tok_sh[i * 64u + lane]);
    }
    EDSX_BLK(15, true)
#undef EDSX_BLK
    // braces (after the tokens: the closing one replaces the last ',')
    if (K) {
        if (lane == 0) {
#pragma unroll
            for (int g = 0; g < K; g++) { text[gstart[g]] = '{'; text[cur[g] - 1] = '}'; }
        }
    } else if (lane < k) { text[gstart_l] = '{'; text[cur_l - 1] = '}'; }
    return run;
}

// ---- id lists: packed prefix sums per block of 64 rows, four groups at a time ----------------------
// Lane `lane` owns rows lane, lane+64, ... (block i = rows i*64 .. i*64+63), exactly as the group ids
// arrive.  The members of a group inside one block are adjacent in the output, so the lanes of a
// block store to consecutive LDS addresses (no bank conflicts beyond the overlap of the k groups).
// Per block one DPP wave scan ranks the rows of ALL groups at once: every lane adds its weight into
// the 8-bit field of its group (4 groups per dword).  The weight is 1 in the blocks whose ids all
// have 3 digits (rank*4 = bytes) and the token length in blocks 0 and 1, whose ids have 1-2 / 2-3
// digits (64 tokens of <= 3 bytes and 35*3 + 29*4 bytes both stay below 256).  Block 15 (ids
// 961..1024, up to 281 bytes) is scanned with 16-bit fields.  Rows that do not exist weigh 0, so
// every block is processed unconditionally: straight-line code whose independent scans interleave.
//   wv    byte i: weight of row i*64+lane (0: no such row)          -- lane constant
//   tokc  [i]: the first four bytes of that row's token "ddd,"      -- lane constant
__device__ __forceinline__ u32 dpp_add(u32 v, u32 moved) { return v + moved; }
template <int CTRL, int ROWMASK> __device__ __forceinline__ u32 dpp_take(u32 v)
{
    return (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWMASK, 0xf, false);   // lanes without a source get 0
}
__device__ __forceinline__ u32 wave_scan_incl(u32 v)
{
    v += dpp_take<0x111, 0xf>(v);        // row_shr:1
    v += dpp_take<0x112, 0xf>(v);        // row_shr:2
    v += dpp_take<0x114, 0xf>(v);        // row_shr:4
    v += dpp_take<0x118, 0xf>(v);        // row_shr:8
    v += dpp_take<0x142, 0xa>(v);        // row_bcast:15 -> rows 1, 3
    v += dpp_take<0x143, 0xc>(v);        // row_bcast:31 -> rows 2, 3
    return v;
}

// Up to four groups (ids 0..3 in `gid`; rows with weight 0 in `wv` are not placed).  Group g's list
// starts at text + run0 + (lists before it); returns the offset behind the last list.
// NB: blocks of 64 rows that can hold rows (16; 4 when S <= 256, where only gid.x / wv.x are looked at).
template <int NB>
__device__ __forceinline__ u32 fast_emit_ids_cols(const uint4& gid, u32 k, uint8_t* text, const u32 (&tokc)[16],
                                                  const uint4& wv, u32 lane, u32 run0)
{
    const u32 tbase = (u32)(uintptr_t)text;   // LDS byte address (low half of the flat address)
    u32 ex[15];                    // exclusive packed prefix of the row inside its block (8-bit fields)
    u32 tot[15];                   // packed block totals (wave-uniform)
    u32 ex15[2], tot15[2];
    // byte totals, 16-bit fields: A[0] = groups 0 (low half) and 2, A[1] = groups 1 and 3
    u32 A[2] = {0, 0};
#define EDSX_A(I)                                                                                 \
    if constexpr (I < NB) {                                                                       \
        const u32 gi = byte_at<I>(gid);                                                           \
        const u32 f = byte_at<I>(wv) << ((gi << 3) & 31u);                                        \
        const u32 inc = wave_scan_incl(f);                                                        \
        ex[I] = inc - f;                                                                          \
        const u32 t = (u32)__builtin_amdgcn_readlane((int)inc, 63);                               \
        tot[I] = t;                                                                               \
        const u32 lo = t & 0x00ff00ffu, hi = (t >> 8) & 0x00ff00ffu;                              \
        A[0] += (I >= 2) ? lo << 2 : lo;                                                          \
        A[1] += (I >= 2) ? hi << 2 : hi;                                                          \
    }
    EDSX_A(0) EDSX_A(1) EDSX_A(2) EDSX_A(3) EDSX_A(4) EDSX_A(5) EDSX_A(6) EDSX_A(7)
    EDSX_A(8) EDSX_A(9) EDSX_A(10) EDSX_A(11) EDSX_A(12) EDSX_A(13) EDSX_A(14)
#undef EDSX_A
    ex15[0] = ex15[1] = tot15[0] = tot15[1] = 0;
    if constexpr (NB == 16) {
        const u32 gi = byte_at<15>(gid);
        const u32 f = byte_at<15>(wv) << (((gi >> 1) & 1u) * 16u);
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const u32 fj = (gi & 1u) == (u32)j ? f : 0u;
            const u32 inc = wave_scan_incl(fj);
            ex15[j] = inc - fj;
            tot15[j] = (u32)__builtin_amdgcn_readlane((int)inc, 63);
            A[j] += tot15[j];
        }
    }
    // group starts, and the packed cursors (same field layout as A)
    u32 C[2] = {0, 0};
    u32 run = run0, gstart[4], gend[4];
#pragma unroll
    for (int g = 0; g < 4; g++) {
        gstart[g] = run;
        if ((u32)g < k) {
            const int j = g & 1, sh = 16 * (g >> 1);
            C[j] |= (run + 1) << sh;
            run += 1 + ((A[j] >> sh) & 0xffffu);
        }
        gend[g] = run;
    }
#define EDSX_CUR(gi) ((((gi & 1u) ? C[1] : C[0]) >> (((gi >> 1) & 1u) * 16u)) & 0xffffu)
#define EDSX_B(I)                                                                                 \
    if constexpr (I < NB) {                                                                       \
        const u32 gi = byte_at<I>(gid);                                                           \
        const u32 rank = (ex[I] >> ((gi << 3) & 31u)) & 0xffu;                                    \
        const u32 off = EDSX_CUR(gi) + ((I >= 2) ? rank << 2 : rank);                             \
        if (byte_at<I>(wv)) {                                                                     \
            const u32 a = tbase + off, t = tokc[I];                                               \
            if (I >= 2) lds_put4(a, t);                                                           \
            else if (I == 0) { lds_put2(a, t); if (lane >= 9u) lds_put1<2>(a, t >> 16); }         \
            else { lds_put2(a, t); lds_put1<2>(a, t >> 16); if (lane >= 35u) lds_put1<3>(a, t >> 24); } \
        }                                                                                         \
        const u32 lo = tot[I] & 0x00ff00ffu, hi = (tot[I] >> 8) & 0x00ff00ffu;                    \
        C[0] += (I >= 2) ? lo << 2 : lo;                                                          \
        C[1] += (I >= 2) ? hi << 2 : hi;                                                          \
    }
    EDSX_B(0) EDSX_B(1) EDSX_B(2) EDSX_B(3) EDSX_B(4) EDSX_B(5) EDSX_B(6) EDSX_B(7)
    EDSX_B(8) EDSX_B(9) EDSX_B(10) EDSX_B(11) EDSX_B(12) EDSX_B(13) EDSX_B(14)
#undef EDSX_B
    if constexpr (NB == 16) {
        const u32 gi = byte_at<15>(gid);
        const u32 pk = (gi & 1u) ? ex15[1] : ex15[0];
        const u32 off = EDSX_CUR(gi) + ((pk >> (((gi >> 1) & 1u) * 16u)) & 0xffffu);
        if (byte_at<15>(wv)) {
            const u32 a = tbase + off;
            lds_put4(a, tokc[15]);
            if (lane >= 39u) lds_put1<4>(a, (u32)',');
        }
    }
#undef EDSX_CUR
    if (lane == 0) {                               // braces last: the closing one replaces the final ','
#pragma unroll
        for (int g = 0; g < 4; g++) if ((u32)g < k) { text[gstart[g]] = '{'; text[gend[g] - 1] = '}'; }
    }
    return run;
}

// More than four groups (k <= 16): four at a time.  The group ids outside the current quartet get
// weight 0 (SWAR: a byte is inside iff (id ^ base) & 0xFC == 0), the ids inside become 0..3.
template <int NB>
__device__ __forceinline__ u32 fast_emit_ids_cols_multi(const uint4& gid, u32 k, uint8_t* text, const u32 (&tokc)[16],
                                                        const uint4& wv, u32 lane)
{
    u32 run = 0;
    for (u32 gb = 0; gb < k; gb += 4) {
        const u32 bb = gb * 0x01010101u;
        uint4 g2, w2;
#define EDSX_Q(C)                                                                                 \
        { const u32 x = gid.C ^ bb; g2.C = x & 0x03030303u; w2.C = wv.C & ~bytes_ne_mask(x & 0xfcfcfcfcu, 0u); }
        EDSX_Q(x) EDSX_Q(y) EDSX_Q(z) EDSX_Q(w)
#undef EDSX_Q
        run = fast_emit_ids_cols<NB>(g2, k - gb < 4u ? k - gb : 4u, text, tokc, w2, lane, run);
    }
    return run;
}

template <int NB>
__global__ void __launch_bounds__(256, 4) k_emit_variant_fast(FastParams p)
{
    __shared__ __attribute__((aligned(16))) u64 tok_sh[1024];
    __shared__ __attribute__((aligned(16))) uint8_t stage_sh[4][FAST_STAGE];
    const MsaView& mv = p.mv;
    if (mv.hdr->status) return;
    const u32 lane = threadIdx.x & 63, wv = uniform32(threadIdx.x >> 6);
    for (u32 r = threadIdx.x; r < 1024; r += 256) {
        u32 id = r + 1, nd = ndigits(id);
        u64 t = 0;
        u32 v = id;
        for (int i = (int)nd - 1; i >= 0; i--) { t |= (u64)('0' + v % 10u) << (8 * i); v /= 10u; }
        t |= (u64)',' << (8 * nd);
        tok_sh[r] = t | ((u64)(nd + 1) << 56);
    }
    __syncthreads();
    uint8_t* stage = stage_sh[wv];
    // lane constants of the packed-scan emitter: tokens and weights of rows lane, lane+64, ...
    u32 tokc[16];
    uint4 wgt;
    {
        uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const u32 r = (u32)i * 64u + lane;
            tokc[i] = (u32)tok_sh[r];
            const u32 tl = (u32)(tok_sh[r] >> 56);
            if (r < mv.S) w[i >> 2] |= ((i < 2 || i == 15) ? tl : 1u) << ((i & 3) * 8);
        }
        wgt = make_uint4(w[0], w[1], w[2], w[3]);
    }
    const u64 nseg = *p.nseg_ptr;
    const u64 p0 = mv.vbit(0) ? 0 : 1;
    const u64 nvs = nseg > p0 ? (nseg - p0 + 1) / 2 : 0;
    const u64 nw = ((u64)gridDim.x * blockDim.x) >> 6;
    const uint4 vmask = fast_valid_mask(lane, mv.S);
    struct Rec { uint4 gid; u32 rep, chr, k; };
    auto load_rec = [&](u64 v) -> Rec {
        Rec r;
        const uint8_t* rec = p.grec + v * (u64)GREC_BYTES;
        r.gid = *reinterpret_cast<const uint4*>(rec + lane * 16u);
        r.rep = *reinterpret_cast<const uint16_t*>(rec + GREC_REP + lane * 2u);
        r.chr = rec[GREC_CHR + lane];
        r.k = rec[GREC_K];
        return r;
    };
    // Software pipeline over the wave's segments t, t+nw, t+2nw ...: vmcnt retires in issue order, so a
    // wait for a prefetched record also waits for every store issued before it.  The record of
    // segment t+2 is therefore requested first, the id text of segment t is built in LDS (no
    // global traffic), and only then the wave waits — by now the stores of segment t-1 have had the
    // whole build phase to retire — and issues the stores of segment t.
    u64 vi = (u64)blockIdx.x * (blockDim.x >> 6) + uniform32(threadIdx.x >> 6);
    const u64 v0 = vi < nvs ? vi : 0, v1 = vi + nw < nvs ? vi + nw : v0;
    Rec rc = load_rec(v0), rc_n = load_rec(v1);
    u64 meta = uniform64(p.segmeta[2 * v0 + p0]), meta_n = uniform64(p.segmeta[2 * v1 + p0]);
    u64 goff = uniform64(p.seds_len[2 * v0 + p0]), goff_n = uniform64(p.seds_len[2 * v1 + p0]);   // offsets after the scans
    u64 eoff = uniform64(p.eds_len[2 * v0 + p0]), eoff_n = uniform64(p.eds_len[2 * v1 + p0]);
    // nothing may be pending at the loop header, or the wait for it is placed inside the loop
    asm volatile("" :: "v"(rc.gid.x), "v"(rc.gid.y), "v"(rc.gid.z), "v"(rc.gid.w), "v"(rc.rep), "v"(rc.chr), "v"(rc.k),
                       "v"(rc_n.gid.x), "v"(rc_n.gid.y), "v"(rc_n.gid.z), "v"(rc_n.gid.w), "v"(rc_n.rep), "v"(rc_n.chr), "v"(rc_n.k));
    while (vi < nvs) {
        const u64 seg = 2 * vi + p0;
        const u64 v2 = vi + 2 * nw < nvs ? vi + 2 * nw : vi;
        const Rec rc_nn = load_rec(v2);
        const u64 meta_v = p.segmeta[2 * v2 + p0];          // same address in every lane; made scalar
        const u64 goff_v = p.seds_len[2 * v2 + p0];         // only after the wait below
        const u64 eoff_v = p.eds_len[2 * v2 + p0];
        FastGroups G;
        G.gid = rc.gid; G.k = uniform32(rc.k); G.rep = rc.rep; G.key_lo = rc.chr; G.key_hi = 0; G.sumlen = 0; G.len = 0;
        const bool fast = G.k != 0;
        u32 n = 0;
        const u32 sh = (u32)goff & 15u;
        if (fast) {
            uint8_t* text = stage + sh;                    // LDS offset == global offset (mod 16)
            if (G.k <= 4) n = fast_emit_ids_cols<NB>(G.gid, G.k, text, tokc, wgt, lane, 0u);
            else if (G.k <= 16) n = fast_emit_ids_cols_multi<NB>(G.gid, G.k, text, tokc, wgt, lane);
            else n = fast_emit_ids<0>(G.gid, G.k, text, tok_sh, lane);
        }
        // the wait for the prefetched record (and with it for the previous segment's stores) goes here
        asm volatile("" :: "v"(rc_nn.gid.x), "v"(rc_nn.gid.y), "v"(rc_nn.gid.z), "v"(rc_nn.gid.w),
                           "v"(rc_nn.rep), "v"(rc_nn.chr), "v"(rc_nn.k), "v"(meta_v), "v"(goff_v), "v"(eoff_v));
        const u64 meta_nn = uniform64(meta_v), goff_nn = uniform64(goff_v), eoff_nn = uniform64(eoff_v);
        if (fast) {
            // ---- eds: "{" s0 "," s1 ... "}"
            {
                uint8_t* e = p.eds + eoff;
                const u32 ncol = (u32)(meta >> 48) & 0xffu;
                if (ncol == 1) {                          // lane g holds group g's letter (0 = empty string)
                    const u32 c = lane < G.k ? (u32)G.key_lo : 0u;
                    const u64 nz = ballot64(c != 0);
                    const u32 at = 1 + lane + mbcnt(nz);  // '{' + one separator per earlier group + earlier letters
                    if (lane == 0) e[0] = '{';
                    if (lane < G.k) {
                        if (c) e[at] = (uint8_t)c;
                        e[at + (c ? 1 : 0)] = (lane + 1 < G.k) ? ',' : '}';
                    }
                } else if (G.k * ncol <= 64u) {           // lane = (group, column): one load round trip
                    const u32 g = lane / ncol, c = lane - g * ncol;
                    const u32 r = (u32)__shfl((int)G.rep, (int)(g < G.k ? g : 0u), 64);
                    u32 ch = 0;
                    if (g < G.k) {
                        const u64 sl = (meta & META_SCATTER) ? mv.slot(p.seg_start[seg] + c) : (meta & META_SLOT) + c;
                        ch = mv.vc[sl * (u64)mv.Spad + vc_pos(r, mv.Gp)];
                        if (ch == '-' || ch == '\n') ch = 0;
                    }
                    const u64 m = ballot64(ch != 0);
                    if (lane == 0) e[0] = '{';
                    if (ch) e[1 + g + mbcnt(m)] = (uint8_t)ch;
                    if (g < G.k && c == 0) {              // separator after group g's letters
                        const u32 endl = (g + 1) * ncol;
                        const u64 upto = endl >= 64u ? ~0ull : ((1ull << endl) - 1);
                        e[1 + g + (u32)__builtin_popcountll(m & upto)] = (g + 1 < G.k) ? ',' : '}';
                    }
                } else {
                    if (lane == 0) e[0] = '{';
                    u32 eo = 1;
                    for (u32 g = 0; g < G.k; g++) {       // lane = column: the representative row's letters
                        const u32 r = (u32)__builtin_amdgcn_readlane((int)G.rep, (int)g);
                        u32 ch = 0;
                        if (lane < ncol) {
                            const u64 sl = (meta & META_SCATTER) ? mv.slot(p.seg_start[seg] + lane) : (meta & META_SLOT) + lane;
                            ch = mv.vc[sl * (u64)mv.Spad + vc_pos(r, mv.Gp)];
                            if (ch == '-' || ch == '\n') ch = 0;
                        }
                        const u64 m = ballot64(ch != 0);
                        const u32 len = (u32)__builtin_popcountll(m);
                        if (ch) e[eo + mbcnt(m)] = (uint8_t)ch;
                        if (lane == 0) e[eo + len] = (g + 1 < G.k) ? ',' : '}';
                        eo += len + 1;
                    }
                }
            }
            // ---- flush LDS -> HBM: aligned 16-byte blocks, byte stores for the ragged ends
            uint8_t* gdst = p.seds + (goff - sh);          // 16-byte aligned image of `stage`
            const u32 endo = sh + n;
            const u32 body0 = sh ? 16u : 0u, body1 = endo & ~15u;
            if (sh && lane < 16) { u32 o = lane; if (o >= sh && o < endo) gdst[o] = stage[o]; }
            for (u32 o = body0 + lane * 16u; o + 16u <= body1; o += 1024u)
                *reinterpret_cast<uint4*>(gdst + o) = *reinterpret_cast<const uint4*>(stage + o);
            {
                const u32 t0 = body1 > body0 ? body1 : body0;
                if (lane < 16) { u32 o = t0 + lane; if (o < endo) gdst[o] = stage[o]; }
            }
        }
        vi += nw;
        rc = rc_n; meta = meta_n; goff = goff_n; eoff = eoff_n;
        rc_n = rc_nn; meta_n = meta_nn; goff_n = goff_nn; eoff_n = eoff_nn;
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

template <int T, int RPT, bool HOLD, bool LANEROWS, int MINW>
static void launch_k1(const K1Params& p, size_t lds, hipStream_t st)
{
    auto kern = k_scan_extract<T, RPT, HOLD, LANEROWS, MINW>;
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
    idx_tmp_.ensure(2 * sizeof(u64) * ROW_CAP);
    TIMED("k_index_spec", st, hipLaunchKernelGGL(k_index_spec, dim3(512), dim3(256), 0, st, d_msa, (u64)n, dh,
                                                 idx_tmp_.as<u64>(), idx_tmp_.as<u64>() + ROW_CAP, (u64)ROW_CAP));
    TIMED("k_index_check", st, hipLaunchKernelGGL(k_index_check, dim3(1), dim3(1024), 0, st, d_msa, (u64)n, dh,
                                                  idx_tmp_.as<u64>(), idx_tmp_.as<u64>() + ROW_CAP, rows_.as<u64>(), (u64)ROW_CAP));
    TIMED("k_index_rows", st, hipLaunchKernelGGL(k_index_rows, dim3(1), dim3(64), 0, st, d_msa, (u64)n, dh,
                                                 rows_.as<u64>(), (u64)ROW_CAP));
    EDSX_HIP(hipMemcpyAsync(&h_, dh, sizeof(MsaHdr), hipMemcpyDeviceToHost, st));
    EDSX_HIP(hipStreamSynchronize(st));
    if (h_.status) throw FormatError(status_message(h_.status));
    if (h_.S > MAX_ROWS) throw FormatError(status_message(ST_TOO_MANY_ROWS));

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
    scan_tmp_.ensure(8 * ((L + 2) / SCAN_TILE + 2));
    run_start_.ensure(8 * (L + 2));
    if (l) { flag_.ensure(8 * (L + 2)); seg_start_.ensure(8 * (L + 2)); }
    eds_len_.ensure(8 * (L + 2));
    seds_len_.ensure(8 * (L + 2));

    // ---- K1 geometry.  A workgroup of T threads keeps RPT 16-byte chunks per thread in registers:
    // rows per pass = T / CPR, tile width W = 16 * CPR columns.  Two workgroups per CU let one
    // tile's extraction phase overlap the other's load phase.
    //   cfg 0: T=1024 RPT=16  (1 workgroup/CU, widest tile)
    //   cfg 1: T=512  RPT=16  (2 workgroups/CU)
    //   cfg 2: T=1024 RPT=8   (2 workgroups/CU, <= 64 VGPRs)
    static int cfg_env = -1;
    if (cfg_env < 0) { const char* e = getenv("EDSX_K1"); cfg_env = e ? atoi(e) : 1; if (cfg_env < 0 || cfg_env > 2) cfg_env = 1; }
    int cfg = cfg_env;
    const int T = cfg == 1 ? 512 : 1024, RPT = cfg == 2 ? 8 : 16;
    u32 cpr_log2 = 8;
    while (cpr_log2 > 2 && (u64)RPT * (T >> cpr_log2) < S) cpr_log2--;
    const bool hold = (u64)RPT * (T >> cpr_log2) >= S;
    const u64 W = 16ull << cpr_log2;
    const u64 ntiles = (Draw + W - 1) / W;
    const size_t colbuf_bytes = cfg == 0 ? 96 * 1024 : 64 * 1024;
    if (ntiles > 0x7fffffffull) throw FormatError("MSA too large for one launch");
    if (colbuf_bytes < (size_t)S * 8) throw FormatError(status_message(ST_TOO_MANY_ROWS));

    K1Params kp;
    kp.file = d_msa; kp.row_start = rows_.as<u64>(); kp.hdr = dh;
    kp.Vraw = vraw_.as<u64>(); kp.word_slot = wslot_.as<u64>(); kp.vc = vc_.as<uint8_t>();
    kp.vc_cap_cols = vc_cap_cols_; kp.Draw = Draw; kp.lw = lw; kp.S = (u32)S; kp.Spad = Spad; kp.Gp = vc_rows_per_lane((u32)S);
    kp.cpr_log2 = cpr_log2; kp.cap_cols = (u32)(colbuf_bytes / Spad); kp.ntiles = ntiles;
    if (kp.cap_cols == 0) throw FormatError(status_message(ST_TOO_MANY_ROWS));
    launch_timer_begin("k_scan_extract", st);
    const bool lane_rows = hold && RPT == 16 && S <= 1024 && (u32)(T >> cpr_log2) == 4 * kp.Gp;   // thread rows = 16 consecutive vc bytes
    if (cfg == 1) {
        if (lane_rows) launch_k1<512, 16, true, true, 4>(kp, colbuf_bytes, st);
        else if (hold) launch_k1<512, 16, true, false, 4>(kp, colbuf_bytes, st);
        else launch_k1<512, 16, false, false, 4>(kp, colbuf_bytes, st);
    } else if (cfg == 2) {
        if (hold) launch_k1<1024, 8, true, false, 8>(kp, colbuf_bytes, st);
        else launch_k1<1024, 8, false, false, 8>(kp, colbuf_bytes, st);
    } else {
        if (lane_rows) launch_k1<1024, 16, true, true, 4>(kp, colbuf_bytes, st);
        else if (hold) launch_k1<1024, 16, true, false, 4>(kp, colbuf_bytes, st);
        else launch_k1<1024, 16, false, false, 4>(kp, colbuf_bytes, st);
    }
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
    mv_.S = (u32)S; mv_.Spad = Spad; mv_.Gp = vc_rows_per_lane((u32)S);
    seg_start_p_ = seg_start; hseg_p_ = Hseg; segbase_p_ = segbase; nseg_p_ = d_nseg;
    seg_lds_ = (size_t)18 * S + 64 + (S <= HT_MAX_ROWS ? HtLds::BYTES + 16 : 0);
    seg_lds_ = (seg_lds_ + 15) & ~(size_t)15;
    stage_off_ = 0;
    stage_cols_ = 0;
    {                                                     // as many columns of a segment as fit beside that (<= STAGE_COLS)
        const size_t budget = (size_t)150 * 1024;
        if (seg_lds_ + 4 * ((size_t)Spad + 8) <= budget) {
            stage_cols_ = (u32)std::min<size_t>(STAGE_COLS, (budget - seg_lds_) / ((size_t)Spad + 8));
            stage_off_ = (u32)seg_lds_;
            seg_lds_ += (size_t)stage_cols_ * 8 + (size_t)stage_cols_ * Spad;
        }
    }

    const u64 tok_total = token_total((u32)S);
    fast_ = S <= 1024;
    SegParams sp;
    sp.mv = mv_; sp.seg_start = seg_start; sp.nseg_ptr = d_nseg; sp.eds_len = eds_len_.as<u64>();
    sp.seds_len = seds_len_.as<u64>(); sp.tok_total = tok_total; sp.list = nullptr; sp.list_n = nullptr;
    sp.stage_cols = stage_cols_; sp.stage_off = stage_off_;
    EDSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_seg_count),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)seg_lds_));
    if (fast_) {
        segmeta_.ensure(8 * (L + 2));
        // list 1: too wide or mixed segments (k_seg_meta), list 2: those the fast kernel gives up on; together
        // at most all variant segments (<= L/2 + 1)
        slow_list_.ensure(8 * (L + 8));
        fp_.mv = mv_; fp_.seg_start = seg_start; fp_.nseg_ptr = d_nseg; fp_.segmeta = segmeta_.as<u64>();
        fp_.eds_len = eds_len_.as<u64>(); fp_.seds_len = seds_len_.as<u64>();
        fp_.slow_list = slow_list_.as<u64>(); fp_.slow_count = &dh->slow_n;
        fp_.slow_list2 = slow_list_.as<u64>() + (L / 2 + 4); fp_.slow_count2 = &dh->slow_n2;
        fp_.eds = nullptr; fp_.seds = nullptr; fp_.tok_total = tok_total;
        // one record per variant segment; there are at most as many as variant columns
        grec_.ensure(((size_t)vc_cap_cols_ + 2) * 1280);
        fp_.grec = grec_.as<uint8_t>();
        EDSX_HIP(hipMemsetAsync(&dh->slow_n, 0, 2 * sizeof(u64), st));
        TIMED("k_seg_meta", st, hipLaunchKernelGGL(k_seg_meta, dim3(4096), dim3(256), 0, st, fp_));
        if (S <= 256) {                                       // at most four rows per lane: leaner instantiation
            TIMED("k_seg_count_fast", st, hipLaunchKernelGGL(k_seg_count_fast<4>, dim3(persistent_grid(
                      reinterpret_cast<const void*>(k_seg_count_fast<4>), 256, 0)), dim3(256), 0, st, fp_));
        } else {
            TIMED("k_seg_count_fast", st, hipLaunchKernelGGL(k_seg_count_fast<16>, dim3(persistent_grid(
                      reinterpret_cast<const void*>(k_seg_count_fast<16>), 256, 0)), dim3(256), 0, st, fp_));
        }
        sp.list = fp_.slow_list; sp.list_n = fp_.slow_count;
        TIMED("k_seg_count_slow", st, hipLaunchKernelGGL(k_seg_count, dim3(seg_grid()), dim3(GT), seg_lds_, st, sp));
        sp.list = fp_.slow_list2; sp.list_n = fp_.slow_count2;
        TIMED("k_seg_count_slow2", st, hipLaunchKernelGGL(k_seg_count, dim3(seg_grid()), dim3(GT), seg_lds_, st, sp));
    } else {
        TIMED("k_seg_count", st, hipLaunchKernelGGL(k_seg_count, dim3(seg_grid()), dim3(GT), seg_lds_, st, sp));
    }
    TIMED("scan_eds", st, exclusive_scan_u64(eds_len_.as<u64>(), eds_len_.as<u64>(), d_nseg, &dh->E,
                                             scan_tmp_.as<u64>(), st));
    TIMED("scan_seds", st, exclusive_scan_u64(seds_len_.as<u64>(), seds_len_.as<u64>(), d_nseg, &dh->Q,
                                              scan_tmp_.as<u64>(), st));
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

// first / last segment of the planned alignment: type, width and text sizes (for the slab stitch)
MsaPipeline::Edges MsaPipeline::edge_info(hipStream_t st)
{
    if (!planned_) throw ParamError("edge_info needs a planned alignment");
    const u64 nseg = h_.nseg;
    Edges e{};
    e.nseg = nseg;
    u64 ss[2], sl[1], eo[2], so[2], el[1], slo[1], v0 = 0, vl = 0;
    EDSX_HIP(hipMemcpyAsync(ss, seg_start_p_, 16, hipMemcpyDeviceToHost, st));
    EDSX_HIP(hipMemcpyAsync(sl, seg_start_p_ + (nseg - 1), 8, hipMemcpyDeviceToHost, st));
    EDSX_HIP(hipMemcpyAsync(eo, eds_len_.as<u64>(), nseg > 1 ? 16 : 8, hipMemcpyDeviceToHost, st));
    EDSX_HIP(hipMemcpyAsync(so, seds_len_.as<u64>(), nseg > 1 ? 16 : 8, hipMemcpyDeviceToHost, st));
    EDSX_HIP(hipMemcpyAsync(el, eds_len_.as<u64>() + (nseg - 1), 8, hipMemcpyDeviceToHost, st));
    EDSX_HIP(hipMemcpyAsync(slo, seds_len_.as<u64>() + (nseg - 1), 8, hipMemcpyDeviceToHost, st));
    EDSX_HIP(hipMemcpyAsync(&v0, mv_.V, 8, hipMemcpyDeviceToHost, st));
    EDSX_HIP(hipMemcpyAsync(&vl, mv_.V + ((h_.L - 1) >> 6), 8, hipMemcpyDeviceToHost, st));
    EDSX_HIP(hipStreamSynchronize(st));
    e.fvar = v0 & 1;
    e.fcols = (nseg > 1 ? ss[1] : h_.L) - ss[0];
    e.feds = nseg > 1 ? eo[1] : h_.E;
    e.fseds = nseg > 1 ? so[1] : h_.Q;
    e.lvar = (vl >> ((h_.L - 1) & 63)) & 1;
    e.lcols = h_.L - sl[0];
    e.leds = h_.E - el[0];
    e.lseds = h_.Q - slo[0];
    return e;
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
    EDSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_emit_variant),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)seg_lds_));
    if (fast_) {
        TIMED("k_emit_common", st, hipLaunchKernelGGL(k_emit_common, dim3(2048), dim3(256), 0, st, ep));
        FastParams fp = fp_;
        fp.eds = d_eds; fp.seds = d_seds;
        if (h_.S <= 256) {                                    // at most four blocks of 64 rows
            TIMED("k_emit_variant_fast", st, hipLaunchKernelGGL(k_emit_variant_fast<4>, dim3(persistent_grid(
                      reinterpret_cast<const void*>(k_emit_variant_fast<4>), 256, 0)), dim3(256), 0, st, fp));
        } else {
            TIMED("k_emit_variant_fast", st, hipLaunchKernelGGL(k_emit_variant_fast<16>, dim3(persistent_grid(
                      reinterpret_cast<const void*>(k_emit_variant_fast<16>), 256, 0)), dim3(256), 0, st, fp));
        }
        ep.list = fp_.slow_list; ep.list_n = fp_.slow_count;
        TIMED("k_emit_variant_slow", st, hipLaunchKernelGGL(k_emit_variant, dim3(seg_grid()), dim3(GT), seg_lds_, st, ep));
        ep.list = fp_.slow_list2; ep.list_n = fp_.slow_count2;
        TIMED("k_emit_variant_slow2", st, hipLaunchKernelGGL(k_emit_variant, dim3(seg_grid()), dim3(GT), seg_lds_, st, ep));
    } else {
        TIMED("k_emit_common", st, hipLaunchKernelGGL(k_emit_common, dim3(2048), dim3(256), 0, st, ep));
        TIMED("k_emit_variant", st, hipLaunchKernelGGL(k_emit_variant, dim3(seg_grid()), dim3(GT), seg_lds_, st, ep));
    }
    EDSX_HIP(hipGetLastError());
}

} // namespace edsx
