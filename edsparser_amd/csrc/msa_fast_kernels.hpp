// msa_fast_kernels.hpp - wave-per-segment kernels for up to 1024 rows: segment tables, grouping, emitters (K3/K5 fast)
// included by msa_device.hip, which is the one translation unit of these kernels (the wave-level helpers are shared
// between the column scan's fused grouping and the wave-per-segment kernels, and everything inlines).
#pragma once
#include "msa_generic_kernels.hpp"

namespace edsx {

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
        if (mv.S <= 64u && !mixed) {
            // Up to 64 rows: ONE ROW PER LANE instead of sixteen (of which four lanes would own rows).  A lane spells its
            // row's gap-stripped string into LDS (64 bytes per lane), strings with the same length and 32-bit hash are
            // compared with the first such row's string dword by dword (exact for every byte value), the groups come out in
            // the order of their first rows.  The column-by-column refinement below costs a pass over the columns per
            // distinct string - with few rows nearly every row is one (configs[1], 64 x 10 Mb: 0.245 -> see DESIGN).
            const bool act = lane < mv.S;
            u32* const mine32 = reinterpret_cast<u32*>(strs + lane * 64u);
#pragma unroll
            for (int i = 0; i < 16; i += 4) *reinterpret_cast<uint4*>(mine32 + i) = make_uint4(0, 0, 0, 0);
            u32 hsh = 2166136261u, len = 0, nul = 0, nl = 0;
            for (u32 c = 0; c < ncol; c++) {
                const u32 ch = act ? (u32)cell_ptr(c)[lane] : (u32)'-';
                nul |= ch == 0 ? 1u : 0u;
                const bool isnl = ch == '\n';
                nl |= isnl ? 1u : 0u;
                if (ch != '-' && !isnl) { strs[lane * 64u + len] = (uint8_t)ch; hsh = (hsh ^ ch) * 16777619u; len++; }
            }
            if (ballot64(act && nul)) return 0;                   // '\0' ends a row's string (msa_transforms.cpp:282): generic kernels
            if (CHECK_NL && ballot64(act && nl)) saw_nl = 1;
            u32 mygid = 0, rep = 0, k = 0, sum = 0;
            u64 todo = ballot64(act);
            while (todo) {
                const int leader = __builtin_ctzll(todo);
                const u32 lh = (u32)__builtin_amdgcn_readlane((int)hsh, leader), ll = (u32)__builtin_amdgcn_readlane((int)len, leader);
                bool same = ((todo >> lane) & 1ull) && hsh == lh && len == ll;
                const u32* lead32 = reinterpret_cast<const u32*>(strs + (u32)leader * 64u);
                for (u32 i = 0; i < (ll + 3u) / 4u; i++) same = same && mine32[i] == lead32[i];      // (zero padding behind the strings)
                const u64 m = ballot64(same);
                if (same) mygid = k;
                if (lane == k) rep = (u32)leader;
                k++; sum += ll;
                todo &= ~m;
            }
            uint32_t gb[4] = {0, 0, 0, 0};                        // group ids of rows 16 l .. 16 l + 15 for the lanes that own a record dword
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const u32 v = (u32)__shfl((int)mygid, (int)((lane * 16u + (u32)i) & 63u), 64);
                gb[i >> 2] |= (v & 0xffu) << ((i & 3) * 8);
            }
            G.gid = make_uint4(gb[0], gb[1], gb[2], gb[3]);
            G.k = k; G.sumlen = sum; G.rep = rep;
            return 1;
        }
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


} // namespace edsx
