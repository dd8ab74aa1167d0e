// msa_rowloop_kernels.hpp - wave-per-segment kernels for ANY number of rows (alignments of more than 1024 sequences)
// included by msa_device.hip, which is the one translation unit of the MSA kernels.
//
// The wave-per-segment kernels of msa_fast_kernels.hpp keep a segment's rows in registers (16 per lane, at most 1024).
// Here a wave WALKS the rows of its segment 64 at a time, one row per lane: a variant segment of up to RL_MAXCOLS pure
// variant columns has strings of at most sixteen letters, so a row's gap-stripped string IS a 128-bit key (first letter
// in byte 0: exact, msa_transforms.cpp:262-293), the distinct strings live one per lane (at most 64, in the order of
// their first rows), and a row block is matched against them with one ballot per string.  That covers nearly every
// variant segment of an EDS (runs of 1..16 columns); wider or mixed segments (l-EDS), more than 64 strings and rows with a NUL
// byte go on the work list of the generic workgroup-per-segment kernels (msa_generic_kernels.hpp), which were the only
// path for more than 1024 rows until round 3 (2000 rows x 2 M columns: 490 GB/s).
//   record of variant segment vi (rl_rec + vi * stride):  u32 k | per string g < 64: its key (2 x u64) at 16 + 16 g, the
//   .seds bytes of its id list (u32) at 1040 + 4 g | group id of every row (u8) from RL_GID on
#pragma once
#include "msa_fast_kernels.hpp"

namespace edsx {

constexpr u32 RL_MAXCOLS = 16, RL_KMAX = 64, RL_UNROLL = 2;
constexpr u32 RL_KEYS = 16, RL_TOT = RL_KEYS + 16 * RL_KMAX, RL_GID = 1344;
__host__ __device__ inline u64 rl_stride_of(u32 S) { return ((u64)RL_GID + S + 16u + 63u) & ~(u64)63u; }   // (+16: a lane's last 16-byte store of group ids)

// letters of an exact key (non-zero bytes from byte 0 up)
__device__ __forceinline__ u32 rl_key_len(u64 key) { return key ? (71u - (u32)__builtin_clzll(key)) / 8u : 0u; }

// "<id>," lengths of the 64 rows r0 .. r0+63 (ids r0+1 ..): at most two lengths, tlA for the lanes of maskA and tlA + 1
// for the others; the first id of the block decides on the scalar unit, one compare per lane
__device__ __forceinline__ void rl_token_lengths(u32 r0, u32 lane, bool valid, u32& tlA, u64& maskA)
{
    const u32 first = r0 + 1;                              // (uniform)
    u32 p10 = 10, d = 1;
    while (first >= p10 && d < 9) { p10 *= 10; d++; }     // p10 = the first id with one digit more
    tlA = d + 1;
    maskA = ballot64(valid && r0 + lane + 1 < p10);
}

// sizes: common segments (a thread each), then a wave per variant segment
__global__ void __launch_bounds__(256) k_rl_count(RlParams p)
{
    const MsaView& mv = p.mv;
    if (mv.hdr->status) return;                           // vc overflow: the host grows vc and replans
    const u64 nseg = *p.nseg_ptr;
    const u64 p0 = mv.vbit(0) ? 0 : 1;                   // variant and common segments alternate
    const u64 nvs = nseg > p0 ? (nseg - p0 + 1) / 2 : 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) mv.hdr->nvs = nvs;
    for (u64 seg = (1 - p0) + 2 * (blockIdx.x * (u64)blockDim.x + threadIdx.x); seg < nseg; seg += 2 * (u64)gridDim.x * blockDim.x) {
        const u64 ncol = p.seg_start[seg + 1] - p.seg_start[seg];
        p.eds_len[seg] = 2 + ncol;
        p.seds_len[seg] = 3;
        p.segmeta[seg] = 0;
        if (common_is_long(ncol)) p.long_list[atomicAdd(p.long_count, 1ull)] = seg;
    }
    const u32 lane = threadIdx.x & 63, S = mv.S;
    // The waves take the variant segments off a counter, one at a time: a segment of sixteen columns and dozens of strings
    // costs fifty times one of a single SNP column, and with a fixed share per wave the kernel ended when the unluckiest
    // wave did (SQ counters, 2000 rows x 2 M columns: 28 % of the wave slots busy on average).
    while (true) {
        u64 vi = 0;
        if (lane == 0) vi = atomicAdd((unsigned long long*)p.next, 1ull);
        vi = readlane64(vi, 0);
        if (vi >= nvs) break;
        const u64 seg = 2 * vi + p0;
        const u64 a = uniform64(p.seg_start[seg]), b = uniform64(p.seg_start[seg + 1]);
        const u32 ncol = (u32)(b - a);
        bool mine = ncol <= RL_MAXCOLS;
        u64 colbase = 0;                                   // lane c: first byte of column c in vc
        if (mine) {
            bool var = true;
            if (lane < ncol) { var = mv.vbit(a + lane); colbase = mv.slot(a + lane) * (u64)mv.Spad; }
            mine = !ballot64(!var);                        // an l-EDS segment with common columns inside: generic
        }
        const bool wide = ncol > 8u;                       // (uniform) strings of more than eight letters are possible
        u64 K = 0, K2 = 0;                                 // lane g: the key of string g (letters 0..7, 8..15)
        u32 TOT = 0, k = 0;                                // ... the bytes of its id list; strings so far (uniform)
        uint8_t* rec = p.rec + vi * p.rec_stride;
        // ---- a single column (most variant segments): SIXTEEN rows per lane, 1024 per step - the column's bytes are
        // compared with a string's letter sixteen at a time (byte-parallel, as in the kernels for up to 1024 rows), the
        // "<id>," bytes of the matching rows are the byte sum of their lengths (v_sad_u8), and the group ids of a lane's
        // rows go out as one 16-byte store.  ~70 instructions per string and 1024 rows instead of ~150 per 64 rows.
        if (mine && ncol == 1u) {
            const uint8_t* col0 = mv.vc + readlane64(colbase, 0);
            for (u32 r0 = 0; r0 < S; r0 += 1024) {
                const u32 nhere = S - r0 < 1024u ? S - r0 : 1024u;
                const uint4 vmask = fast_valid_mask(lane, nhere);
                const u32 off = r0 + lane * 16u;
                const uint4 raw = load16u(col0 + (off + 16u <= mv.Spad ? off : mv.Spad - 16u));
                if (ballot64(any_nul(raw, vmask))) { mine = false; break; }          // '\0' ends a row's string: generic
                u32 nl_dummy = 0;
                const uint4 col = normalise_col<true>(raw, vmask, nl_dummy);        // '-' (and a stray newline) -> 0: no letter
                // "<id>," lengths of this lane's sixteen rows, a byte each
                uint4 tlv;
                {
                    const u32 id0 = off + 1u;
                    u32 p10 = 10, d = 1;
                    while (id0 >= p10 && d < 9) { p10 *= 10; d++; }                  // p10: the first id with one digit more
                    uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        const u32 id = id0 + (u32)i;
                        const u32 tl = d + 1u + (id >= p10 ? 1u : 0u) + (id >= p10 * 10u ? 1u : 0u);   // (sixteen ids cross at most two powers of ten)
                        w[i >> 2] |= tl << ((i & 3) * 8);
                    }
                    tlv = make_uint4(w[0], w[1], w[2], w[3]);
                }
                uint4 rm = vmask, gidv = make_uint4(0, 0, 0, 0);
                auto take = [&](u32 letter, u32 g) {                                 // the remaining rows that spell `letter` join string g
                    uint4 eq = bytes_eq_mask(col, letter * 0x01010101u);
                    eq.x &= rm.x; eq.y &= rm.y; eq.z &= rm.z; eq.w &= rm.w;
                    u32 bytes = __builtin_amdgcn_sad_u8(eq.x & tlv.x, 0u, 0u) + __builtin_amdgcn_sad_u8(eq.y & tlv.y, 0u, 0u) +
                                __builtin_amdgcn_sad_u8(eq.z & tlv.z, 0u, 0u) + __builtin_amdgcn_sad_u8(eq.w & tlv.w, 0u, 0u);
                    bytes = (u32)__builtin_amdgcn_readlane((int)wave_scan_incl(bytes), 63);
                    if (lane == g) TOT += bytes;
                    const u32 gg = g * 0x01010101u;
                    gidv.x = (gidv.x & ~eq.x) | (eq.x & gg); gidv.y = (gidv.y & ~eq.y) | (eq.y & gg);
                    gidv.z = (gidv.z & ~eq.z) | (eq.z & gg); gidv.w = (gidv.w & ~eq.w) | (eq.w & gg);
                    rm.x &= ~eq.x; rm.y &= ~eq.y; rm.z &= ~eq.z; rm.w &= ~eq.w;
                };
                for (u32 g = 0; g < k; g++) take((u32)__builtin_amdgcn_readlane((int)(u32)K, (int)g), g);
                int leader;
                u32 i0;
                while (first_remaining(rm, leader, i0)) {                             // new strings, in the order of their first rows
                    if (k == RL_KMAX) { mine = false; break; }
                    const u32 letter = leader_byte(col, leader, i0);
                    if (lane == k) { K = letter; K2 = 0; TOT = 0; }
                    take(letter, k);
                    k++;
                }
                if (!mine) break;
                if (lane * 16u < nhere) *reinterpret_cast<uint4*>(rec + RL_GID + off) = gidv;
            }
        } else
        // RL_UNROLL blocks of 64 rows per step: the column bytes of all of them are requested before the first block is
        // matched (four blocks per step measured no faster than one: the kernel is bound by its instructions)
        for (u32 r0 = 0; r0 < S && mine; r0 += 64 * RL_UNROLL) {
            u64 key[RL_UNROLL], key2[RL_UNROLL];
            u32 nul = 0;
#pragma unroll
            for (int u = 0; u < (int)RL_UNROLL; u++) {
                const u32 r = r0 + 64u * (u32)u + lane;
                const bool valid = r < S;
                u32 len = 0;
                key[u] = 0; key2[u] = 0;
                // (a loop of exactly ncol steps with the column's base read from lane c: sixteen unrolled steps behind a
                // scalar branch each cost more than the one or two columns most segments have - 2.0 vs 1.3 ms)
                for (u32 c = 0; c < ncol; c++) {
                    const u64 cb = readlane64(colbase, (int)c);
                    const u32 ch = valid ? (u32)mv.vc[cb + r] : (u32)'-';
                    nul |= ch == 0 ? 1u : 0u;
                    const bool keep = ch != '-' && ch != '\n';
                    const u64 put = keep ? (u64)ch << (8u * (len & 7u)) : 0ull;
                    if (!wide || len < 8u) key[u] |= put; else key2[u] |= put;
                    len += keep ? 1u : 0u;
                }
            }
            if (ballot64(nul != 0)) { mine = false; break; }           // '\0' ends a row's string (msa_transforms.cpp:282): generic
#pragma unroll
            for (int u = 0; u < (int)RL_UNROLL; u++) {
                const u32 r = r0 + 64u * (u32)u + lane;
                const bool valid = r < S;
                if (r0 + 64u * (u32)u >= S || !mine) break;             // (uniform)
                u32 tlA;
                u64 maskA;
                rl_token_lengths(r0 + 64u * (u32)u, lane, valid, tlA, maskA);
                u32 gid = 0;
                u64 todo = ballot64(valid);
                for (u32 g = 0; g < k && todo; g++) {                       // the strings known so far
                    const u64 kg = readlane64(K, (int)g), kg2 = wide ? readlane64(K2, (int)g) : 0ull;
                    const u64 m = ballot64(valid && key[u] == kg && (!wide || key2[u] == kg2)) & todo;
                    if ((m >> lane) & 1ull) gid = g;
                    if (lane == g) TOT += (u32)__builtin_popcountll(m & maskA) * tlA + (u32)__builtin_popcountll(m & ~maskA) * (tlA + 1u);
                    todo &= ~m;
                }
                while (todo) {                                              // new strings, in the order of their first rows
                    if (k == RL_KMAX) { mine = false; break; }
                    const int leader = __builtin_ctzll(todo);
                    const u64 nk = readlane64(key[u], leader), nk2 = wide ? readlane64(key2[u], leader) : 0ull;
                    const u64 m = ballot64(valid && key[u] == nk && (!wide || key2[u] == nk2)) & todo;
                    if ((m >> lane) & 1ull) gid = k;
                    if (lane == k) { K = nk; K2 = nk2; TOT = (u32)__builtin_popcountll(m & maskA) * tlA + (u32)__builtin_popcountll(m & ~maskA) * (tlA + 1u); }
                    k++;
                    todo &= ~m;
                }
                if (!mine) break;
                if (valid) rec[RL_GID + r] = (uint8_t)gid;
            }
        }
        if (!mine) {
            if (lane == 0) { p.slow_list[atomicAdd(p.slow_count, 1ull)] = seg; p.segmeta[seg] = 0; }
            continue;
        }
        u32 sum = lane < k ? rl_key_len(K) + rl_key_len(K2) : 0u;
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        if (lane < k) {
            *reinterpret_cast<u64*>(rec + RL_KEYS + 16u * lane) = K;
            *reinterpret_cast<u64*>(rec + RL_KEYS + 16u * lane + 8u) = K2;
            *reinterpret_cast<u32*>(rec + RL_TOT + 4u * lane) = TOT;
        }
        if (lane == 0) {
            *reinterpret_cast<u32*>(rec) = k;
            p.eds_len[seg] = 2 + (u64)(k - 1) + sum;
            p.seds_len[seg] = (u64)k + p.tok_total;
            p.segmeta[seg] = META_REC | vi;
        }
    }
}

// text of the variant segments k_rl_count kept (msa_transforms.cpp:297-317)
__global__ void __launch_bounds__(256) k_rl_emit(RlParams p)
{
    const MsaView& mv = p.mv;
    if (mv.hdr->status) return;
    const u64 nseg = *p.nseg_ptr;
    const u64 p0 = mv.vbit(0) ? 0 : 1;
    const u64 nvs = nseg > p0 ? (nseg - p0 + 1) / 2 : 0;
    const u32 lane = threadIdx.x & 63, S = mv.S;
    // (a fixed share of the segments per wave here: taking them off a counter as k_rl_count does measured slower - 1.28 vs
    // 1.0 ms - neighbouring segments write neighbouring text)
    const u64 wave = (blockIdx.x * (u64)blockDim.x + threadIdx.x) >> 6, nwaves = ((u64)gridDim.x * blockDim.x) >> 6;
    for (u64 vi = wave; vi < nvs; vi += nwaves) {
        const u64 seg = 2 * vi + p0;
        if (!(uniform64(p.segmeta[seg]) & META_REC)) continue;          // a segment of the generic kernels
        const uint8_t* rec = p.rec + vi * p.rec_stride;
        const u32 k = uniform32(*reinterpret_cast<const u32*>(rec));
        u64 K = 0, K2 = 0;
        u32 TOT = 0;
        if (lane < k) {
            K = *reinterpret_cast<const u64*>(rec + RL_KEYS + 16u * lane); K2 = *reinterpret_cast<const u64*>(rec + RL_KEYS + 16u * lane + 8u);
            TOT = *reinterpret_cast<const u32*>(rec + RL_TOT + 4u * lane);
        }
        uint8_t* eds = p.eds + uniform64(p.eds_len[seg]);
        uint8_t* seds = p.seds + uniform64(p.seds_len[seg]);
        // ---- .eds: "{" s0 "," s1 ... "}"
        {
            const u32 len = rl_key_len(K) + rl_key_len(K2), mine = lane < k ? len + 1u : 0u;
            u32 incl = mine;
            for (int o = 1; o < 64; o <<= 1) { const u32 x = __shfl_up(incl, o, 64); if (lane >= (u32)o) incl += x; }
            const u32 at = 1u + incl - mine;
            if (lane == 0) eds[0] = '{';
            if (lane < k) {
                for (u32 i = 0; i < len; i++) eds[at + i] = (uint8_t)((i < 8u ? K : K2) >> (8u * (i & 7u)));
                eds[at + len] = lane + 1u < k ? ',' : '}';
            }
        }
        // ---- .seds: "{" ids of string 0 "}" "{" ... ; lane g keeps the write cursor of string g
        u32 cursor;
        {
            const u32 mine = lane < k ? TOT + 1u : 0u;                  // "{" + the id list (its last ',' becomes "}")
            u32 incl = mine;
            for (int o = 1; o < 64; o <<= 1) { const u32 x = __shfl_up(incl, o, 64); if (lane >= (u32)o) incl += x; }
            cursor = incl - mine;
            if (lane < k) seds[cursor] = '{';
            cursor += 1u;
        }
        for (u32 r0 = 0; r0 < S; r0 += 64 * RL_UNROLL) {
            u32 gidv[RL_UNROLL];
#pragma unroll
            for (int u = 0; u < (int)RL_UNROLL; u++) {                   // the group ids of RL_UNROLL blocks in one round trip
                const u32 r = r0 + 64u * (u32)u + lane;
                gidv[u] = r < S ? (u32)rec[RL_GID + r] : 0xffffu;
            }
#pragma unroll
            for (int u = 0; u < (int)RL_UNROLL; u++) {
                const u32 r = r0 + 64u * (u32)u + lane;
                const bool valid = r < S;
                if (r0 + 64u * (u32)u >= S) break;                         // (uniform)
                const u32 gid = gidv[u];
                u32 tlA;
                u64 maskA;
                rl_token_lengths(r0 + 64u * (u32)u, lane, valid, tlA, maskA);
                const u32 tl = ((maskA >> lane) & 1ull) ? tlA : tlA + 1u;
                u32 myoff = 0;
                u64 todo = ballot64(valid);
                while (todo) {                                              // one step per distinct string of the block
                    const int leader = __builtin_ctzll(todo);
                    const u32 g0 = (u32)__builtin_amdgcn_readlane((int)gid, leader);
                    const u64 m = ballot64(valid && gid == g0);
                    const u32 start = (u32)__builtin_amdgcn_readlane((int)cursor, (int)g0);
                    if ((m >> lane) & 1ull) myoff = start + mbcnt(m & maskA) * tlA + mbcnt(m & ~maskA) * (tlA + 1u);
                    if (lane == g0) cursor = start + (u32)__builtin_popcountll(m & maskA) * tlA + (u32)__builtin_popcountll(m & ~maskA) * (tlA + 1u);
                    todo &= ~m;
                }
                if (valid) {                                                // "<id>," (ids of up to seven digits): built in a register,
                    u64 tok = (u64)',' << (8u * (tl - 1u));                 // stored as 4 + 1 / 2 / 4 bytes or 2 + 1
                    u32 v = r + 1;
                    for (int i = (int)tl - 2; i >= 0; i--) { tok |= (u64)('0' + v % 10u) << (8 * i); v /= 10u; }
                    uint8_t* dst = seds + myoff;
                    if (tl >= 4u) {
                        store_small<4>(dst, tok);
                        if (tl == 5u) dst[4] = (uint8_t)(tok >> 32);
                        else if (tl == 6u) store_small<2>(dst + 4, tok >> 32);
                        else if (tl >= 7u) store_small<4>(dst + tl - 4u, tok >> (8u * (tl - 4u)));
                    } else {
                        store_small<2>(dst, tok);
                        if (tl == 3u) dst[2] = (uint8_t)(tok >> 16);
                    }
                }
            }
        }
        // every id has landed before the closing braces overwrite the last ',' of their lists
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane < k) seds[cursor - 1] = '}';
    }
}

} // namespace edsx
