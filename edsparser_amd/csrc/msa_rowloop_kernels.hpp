// msa_rowloop_kernels.hpp - wave-per-segment kernels for ANY number of rows (alignments of more than 1024 sequences)
// included by msa_device.hip, which is the one translation unit of the MSA kernels.
//
// The wave-per-segment kernels of msa_fast_kernels.hpp keep a segment's rows in registers (16 per lane, at most 1024).
// Here a wave WALKS the rows of its segment 64 at a time, one row per lane: a variant segment of up to RL_MAXCOLS pure
// variant columns has strings of at most eight letters, so a row's gap-stripped string IS a 64-bit key (first letter in
// byte 0: exact, msa_transforms.cpp:262-293), the distinct strings live one per lane (at most 64, in the order of their
// first rows), and a row block is matched against them with one ballot per string.  That covers nearly every variant
// segment of an EDS (runs of 1..8 columns); wider or mixed segments (l-EDS), more than 64 strings and rows with a NUL
// byte go on the work list of the generic workgroup-per-segment kernels (msa_generic_kernels.hpp), which were the only
// path for more than 1024 rows until round 3 (2000 rows x 2 M columns: 490 GB/s).
//   record of variant segment vi (rl_rec + vi * stride):  u32 k | per string g < 64: u64 key at 16 + 8 g, u32 .seds
//   bytes of its id list at 528 + 4 g | group id of every row (u8) from RL_GID on
#pragma once
#include "msa_fast_kernels.hpp"

namespace edsx {

constexpr u32 RL_MAXCOLS = 8, RL_KMAX = 64;
constexpr u32 RL_KEYS = 16, RL_TOT = RL_KEYS + 8 * RL_KMAX, RL_GID = 800;
__host__ __device__ inline u64 rl_stride_of(u32 S) { return ((u64)RL_GID + S + 63u) & ~(u64)63u; }

__device__ __forceinline__ u64 readlane64(u64 v, int lane)
{
    return ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(v >> 32), lane) << 32) | (u32)__builtin_amdgcn_readlane((int)(u32)v, lane);
}
// letters of an exact key (non-zero bytes from byte 0 up)
__device__ __forceinline__ u32 rl_key_len(u64 key) { return key ? (71u - (u32)__builtin_clzll(key)) / 8u : 0u; }

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
    const u64 wave = (blockIdx.x * (u64)blockDim.x + threadIdx.x) >> 6, nwaves = ((u64)gridDim.x * blockDim.x) >> 6;
    for (u64 vi = wave; vi < nvs; vi += nwaves) {
        const u64 seg = 2 * vi + p0;
        const u64 a = uniform64(p.seg_start[seg]), b = uniform64(p.seg_start[seg + 1]);
        const u32 ncol = (u32)(b - a);
        bool mine = ncol <= RL_MAXCOLS;
        u64 base[RL_MAXCOLS];                              // (uniform) first byte of every column in vc
        if (mine) {
            u64 sl = 0;
            bool var = true;
            if (lane < ncol) { var = mv.vbit(a + lane); sl = mv.slot(a + lane); }
            mine = !ballot64(!var);                        // an l-EDS segment with common columns inside: generic
#pragma unroll
            for (int c = 0; c < (int)RL_MAXCOLS; c++) base[c] = readlane64(sl, c) * (u64)mv.Spad;
        }
        u64 K = 0;                                         // lane g: the key of string g
        u32 TOT = 0, k = 0;                                // ... the bytes of its id list; strings so far (uniform)
        uint8_t* rec = p.rec + vi * p.rec_stride;
        for (u32 r0 = 0; r0 < S && mine; r0 += 64) {
            const u32 r = r0 + lane;
            const bool valid = r < S;
            u64 key = 0;
            u32 len = 0, nul = 0;
#pragma unroll
            for (int c = 0; c < (int)RL_MAXCOLS; c++) {
                if (c < (int)ncol) {
                    const u32 ch = valid ? (u32)mv.vc[base[c] + r] : (u32)'-';
                    nul |= ch == 0 ? 1u : 0u;
                    const bool keep = ch != '-' && ch != '\n';
                    key |= keep ? (u64)ch << (8u * len) : 0ull;
                    len += keep ? 1u : 0u;
                }
            }
            if (ballot64(nul != 0)) { mine = false; break; }           // '\0' ends a row's string (msa_transforms.cpp:282): generic
            const u32 tl = ndigits(r + 1) + 1;                           // "<id>,"
            const u32 tlA = (u32)__builtin_amdgcn_readfirstlane((int)tl);   // at most two token lengths in 64 consecutive ids
            const u64 maskA = ballot64(valid && tl == tlA);
            u32 gid = 0;
            u64 todo = ballot64(valid);
            for (u32 g = 0; g < k && todo; g++) {                       // the strings known so far
                const u64 kg = readlane64(K, (int)g);
                const u64 m = ballot64(valid && key == kg) & todo;
                if ((m >> lane) & 1ull) gid = g;
                if (lane == g) TOT += (u32)__builtin_popcountll(m & maskA) * tlA + (u32)__builtin_popcountll(m & ~maskA) * (tlA + 1u);
                todo &= ~m;
            }
            while (todo) {                                              // new strings, in the order of their first rows
                if (k == RL_KMAX) { mine = false; break; }
                const int leader = __builtin_ctzll(todo);
                const u64 nk = readlane64(key, leader);
                const u64 m = ballot64(valid && key == nk) & todo;
                if ((m >> lane) & 1ull) gid = k;
                if (lane == k) { K = nk; TOT = (u32)__builtin_popcountll(m & maskA) * tlA + (u32)__builtin_popcountll(m & ~maskA) * (tlA + 1u); }
                k++;
                todo &= ~m;
            }
            if (!mine) break;
            if (valid) rec[RL_GID + r] = (uint8_t)gid;
        }
        if (!mine) {
            if (lane == 0) { p.slow_list[atomicAdd(p.slow_count, 1ull)] = seg; p.segmeta[seg] = 0; }
            continue;
        }
        u32 sum = lane < k ? rl_key_len(K) : 0u;
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        if (lane < k) {
            *reinterpret_cast<u64*>(rec + RL_KEYS + 8u * lane) = K;
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
    const u64 wave = (blockIdx.x * (u64)blockDim.x + threadIdx.x) >> 6, nwaves = ((u64)gridDim.x * blockDim.x) >> 6;
    for (u64 vi = wave; vi < nvs; vi += nwaves) {
        const u64 seg = 2 * vi + p0;
        if (!(uniform64(p.segmeta[seg]) & META_REC)) continue;          // a segment of the generic kernels
        const uint8_t* rec = p.rec + vi * p.rec_stride;
        const u32 k = uniform32(*reinterpret_cast<const u32*>(rec));
        u64 K = 0;
        u32 TOT = 0;
        if (lane < k) { K = *reinterpret_cast<const u64*>(rec + RL_KEYS + 8u * lane); TOT = *reinterpret_cast<const u32*>(rec + RL_TOT + 4u * lane); }
        uint8_t* eds = p.eds + uniform64(p.eds_len[seg]);
        uint8_t* seds = p.seds + uniform64(p.seds_len[seg]);
        // ---- .eds: "{" s0 "," s1 ... "}"
        {
            const u32 len = rl_key_len(K), mine = lane < k ? len + 1u : 0u;
            u32 incl = mine;
            for (int o = 1; o < 64; o <<= 1) { const u32 x = __shfl_up(incl, o, 64); if (lane >= (u32)o) incl += x; }
            const u32 at = 1u + incl - mine;
            if (lane == 0) eds[0] = '{';
            if (lane < k) {
                for (u32 i = 0; i < len; i++) eds[at + i] = (uint8_t)(K >> (8u * i));
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
        for (u32 r0 = 0; r0 < S; r0 += 64) {
            const u32 r = r0 + lane;
            const bool valid = r < S;
            const u32 gid = valid ? (u32)rec[RL_GID + r] : 0xffffu;
            const u32 tl = ndigits(r + 1) + 1;
            const u32 tlA = (u32)__builtin_amdgcn_readfirstlane((int)tl);
            const u64 maskA = ballot64(valid && tl == tlA);
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
            if (valid) {                                                // "<id>," digit by digit (ids of up to seven digits)
                uint8_t* dst = seds + myoff;
                u32 v = r + 1;
                dst[tl - 1] = ',';
                for (int i = (int)tl - 2; i >= 0; i--) { dst[i] = (uint8_t)('0' + v % 10u); v /= 10u; }
            }
        }
        // every id has landed before the closing braces overwrite the last ',' of their lists
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane < k) seds[cursor - 1] = '}';
    }
}

} // namespace edsx
