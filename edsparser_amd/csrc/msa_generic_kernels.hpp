// msa_generic_kernels.hpp - generic workgroup-per-segment grouping / text kernels and the common-segment text (K3, K5)
// included by msa_device.hip, which is the one translation unit of these kernels (the wave-level helpers are shared
// between the column scan's fused grouping and the wave-per-segment kernels, and everything inlines).
#pragma once
#include "msa_wave.hpp"

namespace edsx {

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
constexpr u32 HT_MAX_ROWS = 4096, HT_SIZE = 8192;   // (4096 rows: 72 KB of row tables + 32 KB of table - one workgroup per CU)
constexpr u32 HT_BM_WORDS = HT_MAX_ROWS / 32 + 8;
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
        pre = bm + HT_BM_WORDS;
        flag = pre + HT_BM_WORDS;
    }
    __host__ __device__ static size_t bytes(u32 hsz) { return (size_t)4 * hsz + 4 * (HT_BM_WORDS + HT_BM_WORDS + 8); }
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
        for (u32 i = threadIdx.x; i < HT_BM_WORDS; i += GT) ht.bm[i] = 0;
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
            const u32 nwords = (S + 31) >> 5;                 // <= 128: one wave scans the first-row counts of the words, 64 at a time
            if (threadIdx.x < 64) {
                u32 carry = 0;
                for (u32 w0 = 0; w0 < nwords; w0 += 64) {
                    const u32 w = w0 + threadIdx.x;
                    const u32 c = w < nwords ? (u32)__builtin_popcount(ht.bm[w]) : 0u;
                    const u32 incl = wave_scan_incl(c);
                    if (w < nwords) ht.pre[w] = carry + incl - c;
                    carry += (u32)__builtin_amdgcn_readlane((int)incl, 63);
                }
                if (threadIdx.x == 0) *rep_sh = carry;
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


} // namespace edsx
