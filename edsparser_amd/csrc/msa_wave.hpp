// msa_wave.hpp - status bits, wave64 helpers and the grouping primitives of the wave-per-segment code (MSA -> EDS)
// included by msa_device.hip, which is the one translation unit of these kernels (the wave-level helpers are shared
// between the column scan's fused grouping and the wave-per-segment kernels, and everything inlines).
#pragma once
#include "msa_device.hpp"

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
__device__ __forceinline__ u64 readlane64(u64 v, int lane)
{
    return ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(v >> 32), lane) << 32) | (u32)__builtin_amdgcn_readlane((int)(u32)v, lane);
}

__device__ __forceinline__ u64 ld_relaxed(const u64* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

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

} // namespace edsx
