// merge_device.hip — EDS -> l-EDS merge (LINEAR with sources / CARTESIAN) on gfx950.
//
// Replaces the reference's round loop (src/cpp/lib/transforms/eds_transforms.cpp):
//   is_leds :439-468 + select_independent_merge_pairs :46-107  -> k_should, max-scan, k_select
//   merge_multiple_pairs :120-196 -> EDS::merge_adjacent (eds.cpp:1425-1695) -> k_pair_count / k_pair_fill
//   reconstruct_eds :207-296 (text round trip, an identity)            -> index compaction (scan)
//   EDS::save / save_sources (eds.cpp:600-631, :641-659)               -> k_fin_* kernels
// The reference rebuilds the whole container for every merged pair (quadratic); here a round
// touches each symbol once and strings are materialised once, at the end.
//
// Data layout in HBM
//   chars/str_off     all original strings back to back (device tokeniser k_tok_*, host tokenisers for odd
//                     text; eds.cpp:39-155 rules)
//   entry pool        one entry per string of every symbol that ever existed: leaves are the
//                     original strings; an entry made by a merge points to its (left, right)
//                     parents, carries its length and, for LINEAR, its source set as a bitset of
//                     W 64-bit words (bit 0 = the universal path "0").  A symbol's entries are
//                     contiguous in the pool, in the reference's product order (left outer).
//   symbols           size[i], ent_off[i], len1[i] (length of the only string when size == 1)
//                     for the current round, double buffered.
#include "merge_device.hpp"
#include <chrono>
#include <thread>

#include <algorithm>
#include <cctype>
#include <cstring>

namespace edsx {

constexpr u32 LEAF = 0xffffffffu;
constexpr u64 NO_ERR = ~0ull;

struct SymArrays { u64* size; u64* ent_off; u64* len1; };
struct Pool { u32* left; u32* right; u32* elen; u64* bits; u32 W; };

// ---- per round -------------------------------------------------------------------------------
// should[i] for the pair (i, i+1): eds_transforms.cpp:75-97
__global__ void k_should(SymArrays s, u64 n, u64 l, u64* __restrict__ runmark, u64* __restrict__ any)
{
    for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        bool sh = false;
        if (i + 1 < n) {
            const bool degA = s.size[i] > 1, degB = s.size[i + 1] > 1;
            if (!degA && i > 0 && s.len1[i] < l) sh = true;
            if (!degB && i + 1 < n - 1 && s.len1[i + 1] < l) sh = true;
            if (degA && degB) sh = true;
        }
        // run mark for the max-scan: start of a run of consecutive `should` -> i+1, a gap -> 0 is
        // not enough (a gap must cut the run), so gaps publish their own index with bit 63 set
        u64 prev_sh = 0;
        if (i > 0) {
            const bool degP = s.size[i - 1] > 1, degA = s.size[i] > 1;
            bool p = false;
            if (!degP && i - 1 > 0 && s.len1[i - 1] < l) p = true;
            if (!degA && i < n - 1 && s.len1[i] < l) p = true;
            if (degP && degA) p = true;
            prev_sh = p;
        }
        // value = 2*(position of the latest event)+kind, kind 1 = run start, 0 = gap; monotone in i
        u64 v = 0;
        if (!sh) v = 2 * (i + 1);
        else if (!prev_sh) v = 2 * (i + 1) + 1;
        runmark[i] = v;
        if (sh) *any = 1;
    }
}

// selected pairs: even offsets inside every run of consecutive `should` (:63-66, :99-103 greedy)
__global__ void k_select(const u64* __restrict__ runscan, u64 n, u64* __restrict__ sel, u64* __restrict__ keep)
{
    for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        const u64 v = runscan[i];
        bool s = false;
        if (v & 1) {                                   // inside a run that started at (v>>1)-1
            const u64 start = (v >> 1) - 1;
            s = ((i - start) & 1) == 0;
        }
        sel[i] = s;
    }
}
__global__ void k_keep(const u64* __restrict__ sel, u64 n, u64* __restrict__ keep)
{
    for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x)
        keep[i] = !(i > 0 && sel[i - 1]);
}

// source-set intersection with the universal path 0 (eds.cpp:1481-1500); returns non-empty?
__device__ __forceinline__ bool src_intersect(const u64* a, const u64* b, u32 W, u64* out)
{
    const bool ua = a[0] & 1, ub = b[0] & 1;
    bool any = false;
    for (u32 w = 0; w < W; w++) {
        u64 v;
        if (ua && ub) v = w == 0 ? 1ull : 0ull;
        else if (ua) v = b[w];
        else if (ub) v = a[w];
        else v = a[w] & b[w];
        if (out) out[w] = v;
        any |= v != 0;
    }
    return any;
}

struct RoundParams {
    SymArrays cur, nxt; Pool pool; u64 n; const u64* sel; const u64* keep; const u64* newidx;
    u64* newcnt; const u64* newoff; u64 pool_used; u64* err_pos; u64* err_big; int linear;
};

// survivors of every selected pair (0 for symbols that are copied)
__global__ void k_pair_count(RoundParams p)
{
    for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i < p.n; i += (u64)gridDim.x * blockDim.x) {
        if (!p.keep[i]) continue;
        u64 cnt = 0;
        if (p.sel[i]) {
            const u64 na = p.cur.size[i], nb = p.cur.size[i + 1];
            if (!p.linear) {
                if (na != 0 && nb > 0xffffffffull / na) { atomicMin(p.err_big, i); cnt = 0; }
                else cnt = na * nb;
            } else {
                const u64 a0 = p.cur.ent_off[i], b0 = p.cur.ent_off[i + 1];
                for (u64 x = 0; x < na; x++)
                    for (u64 y = 0; y < nb; y++)
                        cnt += src_intersect(p.pool.bits + (a0 + x) * p.pool.W, p.pool.bits + (b0 + y) * p.pool.W,
                                             p.pool.W, nullptr);
                if (cnt == 0) atomicMin(p.err_pos, i);           // eds.cpp:1513-1519
            }
        }
        p.newcnt[p.newidx[i]] = cnt;
    }
}

// new symbol list + the entries of the merged symbols (product order: left outer, right inner)
__global__ void k_pair_fill(RoundParams p)
{
    for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i < p.n; i += (u64)gridDim.x * blockDim.x) {
        if (!p.keep[i]) continue;
        const u64 j = p.newidx[i];
        if (!p.sel[i]) {
            p.nxt.size[j] = p.cur.size[i]; p.nxt.ent_off[j] = p.cur.ent_off[i]; p.nxt.len1[j] = p.cur.len1[i];
            continue;
        }
        const u64 na = p.cur.size[i], nb = p.cur.size[i + 1];
        const u64 a0 = p.cur.ent_off[i], b0 = p.cur.ent_off[i + 1];
        u64 o = p.pool_used + p.newoff[j];
        const u64 first = o;
        for (u64 x = 0; x < na; x++)
            for (u64 y = 0; y < nb; y++) {
                if (p.linear) {
                    const u64* sa = p.pool.bits + (a0 + x) * p.pool.W;
                    const u64* sb = p.pool.bits + (b0 + y) * p.pool.W;
                    // test first: slot o belongs to the next symbol once this one's survivors are written
                    if (!src_intersect(sa, sb, p.pool.W, nullptr)) continue;
                    src_intersect(sa, sb, p.pool.W, p.pool.bits + o * p.pool.W);
                }
                p.pool.left[o] = (u32)(a0 + x);
                p.pool.right[o] = (u32)(b0 + y);
                p.pool.elen[o] = p.pool.elen[a0 + x] + p.pool.elen[b0 + y];
                o++;
            }
        const u64 cnt = o - first;
        p.nxt.size[j] = cnt; p.nxt.ent_off[j] = first; p.nxt.len1[j] = cnt == 1 ? p.pool.elen[first] : 0;
    }
}

// ---- final text --------------------------------------------------------------------------------
struct FinParams {
    SymArrays sym; Pool pool; u64 n; const u64* cum;     // cum = exclusive scan of sym.size
    u32* fin_ent; uint8_t* fin_flag;                      // per final string: entry, 1 = first | 2 = last | 4 = brackets
    u64* bytes; u64* sbytes;                              // per final string: eds bytes / seds bytes (-> offsets)
    const uint8_t* chars; const u64* str_off;
    uint8_t* out; uint8_t* sout; u64* err; int compact, linear;
    u32* spill = nullptr; u32 spill_depth = 0;            // serial walk after more than FIN_STACK - 2 rounds: a stack per string in HBM
};

__global__ void k_fin_list(FinParams p)
{
    for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i < p.n; i += (u64)gridDim.x * blockDim.x) {
        const u64 k = p.sym.size[i], e0 = p.sym.ent_off[i], t0 = p.cum[i];
        const bool br = !p.compact || k > 1;                   // eds.cpp:613
        for (u64 x = 0; x < k; x++) {
            const u64 t = t0 + x;
            p.fin_ent[t] = (u32)(e0 + x);
            uint8_t fl = (x == 0 ? 1 : 0) | (x + 1 == k ? 2 : 0) | (br ? 4 : 0);
            p.fin_flag[t] = fl;
            p.bytes[t] = (u64)p.pool.elen[e0 + x] + ((x == 0) ? (br ? 1 : 0) : 1) + ((x + 1 == k && br) ? 1 : 0);
            if (p.linear) {
                const u64* b = p.pool.bits + (e0 + x) * p.pool.W;
                u64 sb = 1;                                    // '{' ; every id is followed by ',' or '}'
                for (u32 w = 0; w < p.pool.W; w++) {
                    u64 v = b[w];
                    while (v) { u32 id = w * 64 + __builtin_ctzll(v); v &= v - 1; sb += ndigits(id ? id : 1) + 1; }
                }
                p.sbytes[t] = sb;
            }
        }
    }
}

// ---- final text ---------------------------------------------------------------------------------------
// A string of the result is the left-to-right concatenation of the leaves of its entry's merge tree.  Walking that tree
// with one thread per string is a chain of dependent loads as long as the string has leaves (l = 32 on a 10 % EDS:
// strings of several hundred leaves, 1.8 ms for the longest one alone).  Every entry carries its length, so a node's
// output offset follows from its parent's (right child = offset + length of the left one) and the subtrees can be
// expanded side by side: a workgroup takes a batch of `ns` consecutive strings, keeps (entry, offset) items on a stack
// in LDS, and in every round each thread pops one item - a leaf is copied to its place, an inner entry pushes its two
// children (empty subtrees are dropped, so the items of one level cover disjoint bytes).  The latency per string is its
// tree's depth, not its number of leaves.  Leaves of 256 bytes and more are copied by the whole workgroup.  A stack that
// would overflow (far beyond 256 x depth items) sends the batch to the serial walk below, which also has the only
// depth limit left.
#if defined(EDSX_EXPERIMENTS) && defined(EDSX_FW_STACK)
constexpr u32 FW_STACK = EDSX_FW_STACK;                   // (test builds: see build.py)
#else
constexpr u32 FW_STACK = 4096;
#endif
constexpr u32 FW_LONG = 32, FW_LONG_MIN = 256;

// Depth of an entry's merge tree <= number of rounds that were run (an entry made in round k has children from earlier
// rounds).  Every round merges at least half of each run of mergeable pairs (greedy non-overlapping pairs,
// eds_transforms.cpp:63-66), so trees are normally ~log2(run length) deep; only products that collapse and reopen a run
// round after round make deeper ones.  A merge of up to FIN_STACK - 2 rounds walks with the stack in registers /
// scratch below; after more rounds (the reference allows 10 000, eds_transforms.cpp:335) the host hands every string a
// stack of rounds + 2 entries in HBM (FinParams::spill), so no depth is refused.
#if defined(EDSX_EXPERIMENTS) && defined(EDSX_FIN_STACK)
constexpr int FIN_STACK = EDSX_FIN_STACK;                 // (test builds: a tiny stack sends ordinary inputs through the spill path)
#else
constexpr int FIN_STACK = 96;
#endif

// the serial walk: one thread, one string (fallback of k_fin_write)
__device__ void fin_eds_direct(const FinParams& p, u64 t)
{
    const uint8_t fl = p.fin_flag[t];
    uint8_t* o = p.out + p.bytes[t];
    {   // opening bracket of the symbol's first string, else the separator (kept branch-free:
        // hipcc 7.2 lost the pointer increment when this was written as nested ifs)
        const bool first = (fl & 1) != 0, wr = !first || (fl & 4) != 0;
        if (wr) *o = first ? '{' : ',';
        o += wr ? 1 : 0;
    }
    u32 local[FIN_STACK];
    u32* stack = p.spill ? p.spill + t * (u64)p.spill_depth : local;
    const int cap = p.spill ? (int)p.spill_depth : FIN_STACK;
    int sp = 0;
    stack[sp++] = p.fin_ent[t];
    while (sp) {
        const u32 e = stack[--sp];
        if (p.pool.left[e] == LEAF) {
            const u64 s0 = p.str_off[p.pool.right[e]], s1 = p.str_off[p.pool.right[e] + 1];
            for (u64 c = s0; c < s1; c++) *o++ = p.chars[c];
        } else {
            if (sp + 2 > cap) { *p.err = 1; break; }
            stack[sp++] = p.pool.right[e];
            stack[sp++] = p.pool.left[e];
        }
    }
    if ((fl & 2) && (fl & 4)) *o++ = '}';
}

// n bytes, any alignment, 16 at a time
__device__ __forceinline__ void fin_copy(uint8_t* dst, const uint8_t* src, u64 n)
{
    u64 i = 0;
    for (; i + 16 <= n; i += 16) {
        const uint4 v = load16u(src + i);
        U128u w{v.x, v.y, v.z, v.w};
        __builtin_memcpy(dst + i, &w, 16);
    }
    for (; i < n; i++) dst[i] = src[i];
}

__global__ void __launch_bounds__(256) k_fin_write(FinParams p, u64 nstr, u64 E, u32 ns)
{
    __shared__ u32 s_ent[FW_STACK];
    __shared__ u32 s_dst[FW_STACK];                    // offset from the batch's first byte
    __shared__ u64 l_src[FW_LONG];
    __shared__ u32 l_dst[FW_LONG], l_len[FW_LONG];
    __shared__ u32 top, nlong, ovf;
    const u32 tid = threadIdx.x;
    const u64 nbatch = (nstr + ns - 1) / ns;
    for (u64 bt = blockIdx.x; bt < nbatch; bt += gridDim.x) {
        const u64 t0 = bt * ns, t = t0 + tid;
        const bool owner = tid < ns && t < nstr;
        const u64 g0 = p.bytes[t0], g1 = t0 + ns < nstr ? p.bytes[t0 + ns] : E;
        uint8_t* gout = p.out + g0;
        __syncthreads();                                   // (the previous batch is done with the stack)
        if (tid == 0) { top = 0; nlong = 0; ovf = g1 - g0 > 0xffffffffull ? 1u : 0u; }
        __syncthreads();
        if (owner && !ovf) {
            const uint8_t fl = p.fin_flag[t];
            u32 o = (u32)(p.bytes[t] - g0);
            const bool first = (fl & 1) != 0, wr = !first || (fl & 4) != 0;
            if (wr) gout[o] = first ? '{' : ',';
            o += wr ? 1u : 0u;
            const u32 e = p.fin_ent[t], len = p.pool.elen[e];
            if (len) { const u32 slot = atomicAdd(&top, 1u); s_ent[slot] = e; s_dst[slot] = o; }   // ns <= 256 < FW_STACK
            if ((fl & 2) && (fl & 4)) gout[o + len] = '}';
        }
        __syncthreads();
        while (!ovf) {
            const u32 n = top;
            if (n == 0) break;
            const u32 take = n < 256u ? n : 256u, base = n - take;
            u32 e = 0, d = 0;
            if (tid < take) { e = s_ent[base + tid]; d = s_dst[base + tid]; }
            __syncthreads();                               // every thread has its item and has seen `top`
            if (tid == 0) top = base;
            __syncthreads();
            if (tid < take) {
                const u32 l = p.pool.left[e], r = p.pool.right[e];
                if (l == LEAF) {
                    const u64 s0 = p.str_off[r], s1 = p.str_off[r + 1];
                    if (s1 - s0 >= FW_LONG_MIN) {
                        const u32 slot = atomicAdd(&nlong, 1u);
                        if (slot < FW_LONG) { l_src[slot] = s0; l_dst[slot] = d; l_len[slot] = (u32)(s1 - s0); }
                        else fin_copy(gout + d, p.chars + s0, s1 - s0);
                    } else fin_copy(gout + d, p.chars + s0, s1 - s0);
                } else {
                    const u32 el = p.pool.elen[l], er = p.pool.elen[r];
                    const u32 cnt = (el ? 1u : 0u) + (er ? 1u : 0u);
                    if (cnt) {
                        u32 slot = atomicAdd(&top, cnt);
                        if (slot + cnt > FW_STACK) ovf = 1u;
                        else {
                            if (er) { s_ent[slot] = r; s_dst[slot] = d + el; slot++; }      // left on top: popped first
                            if (el) { s_ent[slot] = l; s_dst[slot] = d; }
                        }
                    }
                }
            }
            __syncthreads();
            const u32 nl = nlong < FW_LONG ? nlong : FW_LONG;
            if (nl) {                                       // the long leaves of this round: the whole workgroup copies
                for (u32 i = 0; i < nl; i++) {
                    const uint8_t* src = p.chars + l_src[i];
                    uint8_t* dst = gout + l_dst[i];
                    const u32 len = l_len[i], full = len & ~15u;
                    for (u32 c = tid * 16u; c < full; c += 4096u) {
                        const uint4 v = load16u(src + c);
                        U128u w{v.x, v.y, v.z, v.w};
                        __builtin_memcpy(dst + c, &w, 16);
                    }
                    if (tid < len - full) dst[full + tid] = src[full + tid];
                }
                __syncthreads();
                if (tid == 0) nlong = 0;
                __syncthreads();
            }
        }
        if (ovf && owner) fin_eds_direct(p, t);            // (rewrites what the stack had placed already: same bytes)
    }
}

// "{id,id,...}" of every string's source set (eds.cpp:641-659), a thread per string
__global__ void k_fin_write_sources(FinParams p, u64 nstr)
{
    for (u64 t = blockIdx.x * (u64)blockDim.x + threadIdx.x; t < nstr; t += (u64)gridDim.x * blockDim.x) {
        uint8_t* so = p.sout + p.sbytes[t];
        *so++ = '{';
        const u64* b = p.pool.bits + (u64)p.fin_ent[t] * p.pool.W;
        for (u32 w = 0; w < p.pool.W; w++) {
            u64 v = b[w];
            while (v) {
                u32 id = w * 64 + __builtin_ctzll(v);
                v &= v - 1;
                u32 nd = ndigits(id ? id : 1);
                u32 x = id;
                for (int d = (int)nd - 1; d >= 0; d--) { so[d] = (uint8_t)('0' + x % 10); x /= 10; }
                so += nd;
                *so++ = ',';
            }
        }
        so[-1] = '}';
    }
}


// ---- device tokenisers ------------------------------------------------------------------------------
// The .eds / .seds text goes to HBM as it is and is tokenised there (eds.cpp:39-155 / :268-355 rules) straight into
// the layout above: no host arrays, no per-array upload.  The kernels only accept text every byte of which they can
// place (no whitespace before the end, braces alternate, no comma outside braces, ids are digits that fit an int,
// no empty source set); anything else raises `bad`, and the host tokenisers - which own the reference's error texts
// and its treatment of odd but legal text - take the input instead.
// Two passes over 4 KB blocks of the text (256 threads x 16 bytes), nothing kept per byte:
//   count   per block: '{', '}', ',' (ids for .seds), starts of bare runs, characters - what a byte IS needs only
//           the byte in front of it
//   scan    of the block counts (three packed u64 arrays, one multi-array pass)
//   fill    the same classification again; block prefix + a scan inside the workgroup give every byte its place in
//           chars / str_off / sym_first (its source set), and the brace depth in front of it - which is where the
//           text is validated: a byte that does not fit its depth raises `bad` (the arrays are sized by the counts
//           either way, so a text that is rejected writes inside them)
struct TokCtl { u64 n, bad, nblk, totA, totB, totC, maxid, pad; };
constexpr u32 TOK_BLOCK = 4096;

__device__ __forceinline__ bool tok_ws(uint8_t c) { return c == ' ' || (c >= 9 && c <= 13); }
__device__ __forceinline__ bool tok_digit(uint8_t c) { return c >= '0' && c <= '9'; }

struct TokSums { u64 a, b, c; };
// exclusive prefix of `mine` over the 256 threads of the workgroup; `total` = sum over all of them
__device__ __forceinline__ TokSums tok_block_scan(const TokSums& mine, TokSums* wsum, TokSums& total)
{
    TokSums inc = mine;
    const u32 lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int o = 1; o < 64; o <<= 1) {
        const u64 ta = __shfl_up(inc.a, o, 64), tb = __shfl_up(inc.b, o, 64), tc = __shfl_up(inc.c, o, 64);
        if (lane >= (u32)o) { inc.a += ta; inc.b += tb; inc.c += tc; }
    }
    __syncthreads();                                          // (wsum of the previous block has been read)
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    TokSums ex{inc.a - mine.a, inc.b - mine.b, inc.c - mine.c};
    total = TokSums{0, 0, 0};
    for (u32 w = 0; w < 4; w++) {
        const TokSums t = wsum[w];
        if (w < wv) { ex.a += t.a; ex.b += t.b; ex.c += t.c; }
        total.a += t.a; total.b += t.b; total.c += t.c;
    }
    return ex;
}
// the 16 bytes of this thread (nb of them exist) and the byte in front of them (0: start of the text)
__device__ __forceinline__ void tok_load(const uint8_t* raw, u64 n, u64 i0, uint8_t (&c)[16], int& nb, uint8_t& prev)
{
    nb = i0 < n ? (n - i0 < 16 ? (int)(n - i0) : 16) : 0;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (nb > 0) v = *reinterpret_cast<const uint4*>(raw + i0);       // (the buffer has 16 bytes of slack behind the text)
    __builtin_memcpy(c, &v, 16);
    prev = (nb > 0 && i0 > 0) ? raw[i0 - 1] : 0;
}

// .eds.  A = '{' | '}' << 32, B = ',' | bare-run starts << 32, C = characters.  Strings open at '{', ',' and at the
// first character of a bare run (eds.cpp:848-878: text outside braces is a symbol of one string), symbols at '{' and there.
template <bool FILL>
__global__ void __launch_bounds__(256) k_tok_eds(const uint8_t* __restrict__ raw, u64 n, u64* __restrict__ A, u64* __restrict__ B,
                                                 u64* __restrict__ C, uint8_t* __restrict__ chars, u64* __restrict__ str_off,
                                                 u64* __restrict__ sym_first, TokCtl* ctl)
{
    __shared__ TokSums wsum[4];
    const u64 nblk = (n + TOK_BLOCK - 1) / TOK_BLOCK;
    bool bad = false;
    for (u64 blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const u64 i0 = blk * TOK_BLOCK + (u64)threadIdx.x * 16;
        uint8_t c[16], prev;
        int nb;
        tok_load(raw, n, i0, c, nb, prev);
        TokSums mine{0, 0, 0};
        uint8_t pv = prev;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            if (k < nb) {
                const uint8_t ch = c[k];
                if (ch == '{') mine.a += 1;
                else if (ch == '}') mine.a += 1ull << 32;
                else if (ch == ',') mine.b += 1;
                else {
                    mine.c += 1;
                    if (i0 + k == 0 || pv == '}') mine.b += 1ull << 32;
                    if (!FILL && tok_ws(ch)) bad = true;
                }
                pv = ch;
            }
        }
        TokSums total;
        TokSums ex = tok_block_scan(mine, wsum, total);
        if constexpr (!FILL) {
            if (threadIdx.x == 0) { A[blk] = total.a; B[blk] = total.b; C[blk] = total.c; }
        } else {
            ex.a += A[blk]; ex.b += B[blk]; ex.c += C[blk];
            u64 o = ex.a & 0xffffffffull, cl = ex.a >> 32, cm = ex.b & 0xffffffffull, br = ex.b >> 32, h = ex.c;
            pv = prev;
#pragma unroll
            for (int k = 0; k < 16; k++) {
                if (k < nb) {
                    const uint8_t ch = c[k];
                    const u64 depth = o - cl;                            // braces open in front of this byte
                    const u64 sidx = o + cm + br;
                    if (ch == '{') { if (depth != 0) bad = true; str_off[sidx] = h; sym_first[o + br] = sidx; o++; }
                    else if (ch == '}') { if (depth != 1) bad = true; cl++; }
                    else if (ch == ',') { if (depth != 1) bad = true; str_off[sidx] = h; cm++; }
                    else {
                        if (depth > 1) bad = true;
                        if (i0 + k == 0 || pv == '}') { str_off[sidx] = h; sym_first[o + br] = sidx; br++; }
                        chars[h] = ch; h++;
                    }
                    pv = ch;
                }
            }
        }
    }
    if (bad) ctl->bad = 1;
    if (FILL && blockIdx.x == 0 && threadIdx.x == 0) {
        const u64 o = ctl->totA & 0xffffffffull, cl = ctl->totA >> 32, cm = ctl->totB & 0xffffffffull, br = ctl->totB >> 32;
        if (o != cl) ctl->bad = 1;                                       // unterminated group
        str_off[o + cm + br] = ctl->totC;
        sym_first[o + br] = o + cm + br;
    }
}

__global__ void k_tok_leaves(const u64* __restrict__ str_off, u64 m, u32* __restrict__ left, u32* __restrict__ right,
                             u32* __restrict__ elen)
{
    for (u64 s = blockIdx.x * (u64)blockDim.x + threadIdx.x; s < m; s += (u64)gridDim.x * blockDim.x) {
        left[s] = LEAF; right[s] = (u32)s; elen[s] = (u32)(str_off[s + 1] - str_off[s]);
    }
}

__global__ void k_tok_syms(const u64* __restrict__ sym_first, const u64* __restrict__ str_off, u64 n0, SymArrays s)
{
    for (u64 k = blockIdx.x * (u64)blockDim.x + threadIdx.x; k < n0; k += (u64)gridDim.x * blockDim.x) {
        const u64 f = sym_first[k], sz = sym_first[k + 1] - f;
        s.size[k] = sz; s.ent_off[k] = f; s.len1[k] = sz == 1 ? str_off[f + 1] - str_off[f] : 0;
    }
}

// .seds: an id is a maximal run of digits; value -> ok?  (std::stoi range, eds.cpp:336-349)
__device__ __forceinline__ bool tok_number(const uint8_t* raw, u64 i, u64 n, u64& val)
{
    val = 0;
    u64 j = i;
    for (; j < n && raw[j] >= '0' && raw[j] <= '9'; j++) {
        val = val * 10 + (raw[j] - '0');
        if (val > 2147483647ull) return false;
    }
    return true;
}

// .seds.  A = '{' | '}' << 32, B = ids (an id starts at a digit that does not follow a digit).  COUNT also finds the
// largest id (the width of the path bitsets); FILL validates and sets bit `id` of the byte's source set.
template <bool FILL>
__global__ void __launch_bounds__(256) k_tok_seds(const uint8_t* __restrict__ raw, u64 n, u64* __restrict__ A, u64* __restrict__ B,
                                                  u64* __restrict__ C, u64* __restrict__ bits, u32 W, TokCtl* ctl)
{
    __shared__ TokSums wsum[4];
    const u64 nblk = (n + TOK_BLOCK - 1) / TOK_BLOCK;
    bool bad = false;
    u64 mx = 0;
    for (u64 blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const u64 i0 = blk * TOK_BLOCK + (u64)threadIdx.x * 16;
        uint8_t c[16], prev;
        int nb;
        tok_load(raw, n, i0, c, nb, prev);
        TokSums mine{0, 0, 0};
        uint8_t pv = prev;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            if (k < nb) {
                const uint8_t ch = c[k];
                if (ch == '{') mine.a += 1;
                else if (ch == '}') mine.a += 1ull << 32;
                else if (tok_digit(ch)) {
                    if (!tok_digit(pv)) {
                        mine.b += 1;
                        if (!FILL) { u64 v; if (!tok_number(raw, i0 + k, n, v)) bad = true; mx = v > mx ? v : mx; }
                    }
                } else if (ch != ',') bad = true;
                pv = ch;
            }
        }
        TokSums total;
        TokSums ex = tok_block_scan(mine, wsum, total);
        if constexpr (!FILL) {
            if (threadIdx.x == 0) { A[blk] = total.a; B[blk] = total.b; C[blk] = 0; }
        } else {
            ex.a += A[blk];
            u64 o = ex.a & 0xffffffffull, cl = ex.a >> 32;
            pv = prev;
#pragma unroll
            for (int k = 0; k < 16; k++) {
                if (k < nb) {
                    const uint8_t ch = c[k];
                    const u64 depth = o - cl;
                    if (ch == '{') { if (depth != 0) bad = true; o++; }
                    else {
                        if (depth != 1) bad = true;
                        if (ch == '}') {
                            u64 q = i0 + k;                              // a set of commas only is empty (eds.cpp:330)
                            while (q > 0 && raw[q - 1] == ',') q--;
                            if (q == 0 || raw[q - 1] == '{') bad = true;
                            cl++;
                        } else if (tok_digit(ch) && !tok_digit(pv) && depth == 1) {
                            u64 v;
                            tok_number(raw, i0 + k, n, v);
                            atomicOr((unsigned long long*)&bits[(o - 1) * W + (v >> 6)], 1ull << (v & 63));
                        }
                    }
                    pv = ch;
                }
            }
        }
    }
    if (bad) ctl->bad = 1;
    if (!FILL && mx) atomicMax((unsigned long long*)&ctl->maxid, (unsigned long long)mx);
    if (FILL && blockIdx.x == 0 && threadIdx.x == 0 && (ctl->totA & 0xffffffffull) != (ctl->totA >> 32)) ctl->bad = 1;
}

// ---- host ---------------------------------------------------------------------------------------
namespace {

std::string strip_ws(const uint8_t* p, size_t n)
{
    std::string s;
    s.reserve(n);
    for (size_t i = 0; i < n; i++) if (!std::isspace(p[i])) s.push_back((char)p[i]);
    return s;
}

std::string normalize(const std::string& in)                 // eds.cpp:831-881
{
    std::string out, run;
    out.reserve(in.size() + 16);
    int depth = 0;
    for (char ch : in) {
        if (ch == '{') {
            if (!run.empty() && depth == 0) { out += '{'; out += run; out += '}'; run.clear(); }
            out += ch; depth++;
        } else if (ch == '}') { out += ch; depth--; }
        else if (depth > 0) out += ch;
        else run += ch;
    }
    if (!run.empty() && depth == 0) { out += '{'; out += run; out += '}'; }
    return out;
}

// ---- chunk-parallel host tokenisers ---------------------------------------------------------------
// .eds / .seds text is a sequence of brace groups that do not nest, so a cut right behind any '}' is a
// point where the sequential tokenisers below are in their initial state.  Large inputs are cut there and
// tokenised by several host threads into flat arrays.  The fast path only accepts well-formed text;
// anything else (nesting, a stray brace, a character that does not belong, an empty source set, a number
// beyond int) makes it give up, and the sequential code then produces the reference's error text.
unsigned tokenizer_threads(size_t n)
{
    static long par_min = -1;
    if (par_min < 0) { const char* e = getenv("EDSX_TOKENIZE_PAR_MIN"); par_min = e ? atol(e) : (1l << 20); }
    if ((long)n < par_min) return 1;
    const unsigned hc = std::thread::hardware_concurrency();
    return std::max(1u, std::min(16u, hc ? hc : 4u));
}
std::vector<size_t> brace_cuts(const uint8_t* p, size_t n, unsigned nt)
{
    std::vector<size_t> cut(nt + 1, n);
    cut[0] = 0;
    for (unsigned t = 1; t < nt; t++) {
        size_t g = (size_t)((unsigned __int128)n * t / nt);
        if (g < cut[t - 1]) g = cut[t - 1];
        const uint8_t* b = g < n ? static_cast<const uint8_t*>(memchr(p + g, '}', n - g)) : nullptr;
        cut[t] = b ? static_cast<size_t>(b - p) + 1 : n;
    }
    return cut;
}
struct EdsPart { std::vector<uint8_t> chars; std::vector<u64> str_end; std::vector<u64> sym_nstr; bool ok = true; };
void tokenize_eds_range(const uint8_t* p, size_t lo, size_t hi, EdsPart& out)
{
    out.chars.reserve(hi - lo);
    int depth = 0;
    bool run_open = false;                                   // a bare run (text outside braces) is being collected
    u64 nstr = 0;
    auto close_symbol = [&] { out.str_end.push_back(out.chars.size()); out.sym_nstr.push_back(nstr + 1); nstr = 0; };
    for (size_t i = lo; i < hi; i++) {
        const uint8_t ch = p[i];
        if (std::isspace(ch)) continue;
        if (ch == '{') {
            if (depth) { out.ok = false; return; }           // nesting
            if (run_open) { close_symbol(); run_open = false; }
            depth = 1;
        } else if (ch == '}') {
            if (!depth) { out.ok = false; return; }          // stray brace
            close_symbol();
            depth = 0;
        } else if (ch == ',') { out.str_end.push_back(out.chars.size()); nstr++; if (!depth) run_open = true; }
        else { out.chars.push_back(ch); if (!depth) run_open = true; }
    }
    if (depth) { out.ok = false; return; }                   // unterminated group
    if (run_open) close_symbol();
}
// fills chars / str_off / sym_first like the sequential tokeniser; false: not well-formed, use that one
bool tokenize_eds_parallel(const uint8_t* p, size_t n, std::vector<uint8_t>& chars, std::vector<u64>& str_off,
                           std::vector<u64>& sym_first)
{
    const unsigned nt = tokenizer_threads(n);
    if (nt < 2) return false;
    const std::vector<size_t> cut = brace_cuts(p, n, nt);
    std::vector<EdsPart> parts(nt);
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; t++) th.emplace_back([&, t] { tokenize_eds_range(p, cut[t], cut[t + 1], parts[t]); });
    for (auto& x : th) x.join();
    size_t nc = 0, ns = 0, ny = 0;
    for (const auto& pt : parts) { if (!pt.ok) return false; nc += pt.chars.size(); ns += pt.str_end.size(); ny += pt.sym_nstr.size(); }
    chars.resize(nc); str_off.assign(ns + 1, 0); sym_first.assign(ny + 1, 0);
    std::vector<size_t> c0(nt + 1, 0), s0(nt + 1, 0), y0(nt + 1, 0);
    for (unsigned t = 0; t < nt; t++) {
        c0[t + 1] = c0[t] + parts[t].chars.size(); s0[t + 1] = s0[t] + parts[t].str_end.size(); y0[t + 1] = y0[t] + parts[t].sym_nstr.size();
    }
    th.clear();
    for (unsigned t = 0; t < nt; t++)
        th.emplace_back([&, t] {
            const EdsPart& pt = parts[t];
            if (!pt.chars.empty()) memcpy(chars.data() + c0[t], pt.chars.data(), pt.chars.size());
            for (size_t i = 0; i < pt.str_end.size(); i++) str_off[s0[t] + i + 1] = c0[t] + pt.str_end[i];
            u64 first = s0[t];
            for (size_t i = 0; i < pt.sym_nstr.size(); i++) { first += pt.sym_nstr[i]; sym_first[y0[t] + i + 1] = first; }
        });
    for (auto& x : th) x.join();
    return true;
}
struct SedsPart { std::vector<int> ids; std::vector<u64> set_end; int maxid = 0; bool ok = true; };
void tokenize_seds_range(const uint8_t* p, size_t lo, size_t hi, SedsPart& out)
{
    int depth = 0;
    bool have = false;
    long long val = 0;
    size_t set_begin = 0;
    auto flush = [&] { if (have) { out.ids.push_back((int)val); out.maxid = std::max(out.maxid, (int)val); } have = false; val = 0; };
    for (size_t i = lo; i < hi; i++) {
        const uint8_t ch = p[i];
        if (std::isspace(ch)) continue;
        if (!depth) {
            if (ch != '{') { out.ok = false; return; }
            depth = 1; set_begin = out.ids.size();
        } else if (ch == '}') {
            flush();
            if (out.ids.size() == set_begin) { out.ok = false; return; }      // empty path set
            out.set_end.push_back(out.ids.size());
            depth = 0;
        } else if (ch == ',') flush();
        else if (ch >= '0' && ch <= '9') {
            val = val * 10 + (ch - '0'); have = true;
            if (val > 2147483647ll) { out.ok = false; return; }               // std::stoi would throw
        } else { out.ok = false; return; }
    }
    if (depth) out.ok = false;
}
bool tokenize_seds_parallel(const uint8_t* p, size_t n, std::vector<SedsPart>& parts, u64& nsets, int& maxid)
{
    const unsigned nt = tokenizer_threads(n);
    if (nt < 2) return false;
    const std::vector<size_t> cut = brace_cuts(p, n, nt);
    parts.assign(nt, SedsPart());
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; t++) th.emplace_back([&, t] { tokenize_seds_range(p, cut[t], cut[t + 1], parts[t]); });
    for (auto& x : th) x.join();
    nsets = 0; maxid = 0;
    for (const auto& pt : parts) { if (!pt.ok) return false; nsets += pt.set_end.size(); maxid = std::max(maxid, pt.maxid); }
    return nsets > 0;
}

void grow_keep(DevBuf& b, size_t used, size_t need, hipStream_t st)
{
    if (need <= b.cap) return;
    DevBuf nb;
    nb.ensure(std::max(need, b.cap * 2));
    if (used) EDSX_HIP(hipMemcpyAsync(nb.ptr, b.ptr, used, hipMemcpyDeviceToDevice, st));
    EDSX_HIP(hipStreamSynchronize(st));
    std::swap(b.ptr, nb.ptr);
    std::swap(b.cap, nb.cap);
}

} // namespace

// Tokenise on the device (kernels above).  false: the text is not plain, or there is none; the caller then runs the
// host tokenisers.  On success chars / str_off / the leaf entries / the round-0 symbol arrays / the source bitsets
// are in place in HBM.
bool MergePipeline::tokenize_device(const uint8_t* eds, size_t eds_n, const uint8_t* seds, size_t seds_n, bool linear,
                                    hipStream_t st, u64& n0, u64& m, u32& W, bool& head_single, bool& tail_single, u64& head_len)
{
    { const char* e = getenv("EDSX_HOST_TOKENIZER"); if (e && atoi(e)) return false; }      // A/B switch for the parity tests
    size_t end = eds_n, send = linear ? seds_n : 0;
    while (end && std::isspace(eds[end - 1])) end--;
    while (send && std::isspace(seds[send - 1])) send--;
    if (end == 0 || end >= 0xfffffff0ull || (linear && (send == 0 || send >= 0xfffffff0ull))) return false;
    const size_t nmax = std::max(end, send);
    if (!device_scratch_fits(nmax + nmax / 64)) return false;   // the raw text + three counters per 4 KB block
    const u64 nblk_max = (nmax + TOK_BLOCK - 1) / TOK_BLOCK;
    d_raw_.ensure(nmax + 16);
    for (DevBuf* b : {&tk_a_, &tk_b_, &tk_c_}) b->ensure(8 * (nblk_max + 2));
    scan_tmp_.ensure(8 * 3 * (nblk_max / SCAN_TILE + 4));
    TokCtl* ctl = ctl_.as<TokCtl>();
    TokCtl h{};
    h.n = end; h.nblk = (end + TOK_BLOCK - 1) / TOK_BLOCK;
    EDSX_HIP(hipMemcpyAsync(ctl, &h, sizeof(h), hipMemcpyHostToDevice, st));
    EDSX_HIP(hipMemcpyAsync(d_raw_.ptr, eds, end, hipMemcpyHostToDevice, st));
    const uint8_t* raw = d_raw_.as<uint8_t>();
    u64 *a = tk_a_.as<u64>(), *b = tk_b_.as<u64>(), *c = tk_c_.as<u64>();
    const unsigned grid = (unsigned)std::min<u64>(h.nblk, 8192);
    hipLaunchKernelGGL(k_tok_eds<false>, dim3(grid), dim3(256), 0, st, raw, (u64)end, a, b, c, (uint8_t*)nullptr, (u64*)nullptr,
                       (u64*)nullptr, ctl);
    {
        ScanSet<3> ss{{a, b, c}, {a, b, c}, {&ctl->totA, &ctl->totB, &ctl->totC}};
        exclusive_scan_multi<3>(ss, &ctl->nblk, scan_tmp_.as<u64>(), st);
    }
    EDSX_HIP(hipMemcpyAsync(&h, ctl, sizeof(h), hipMemcpyDeviceToHost, st));
    EDSX_HIP(hipStreamSynchronize(st));
    if (h.bad) return false;
    const u64 opens = h.totA & 0xffffffffull, commas = h.totB & 0xffffffffull, bare = h.totB >> 32;
    const u64 nchars = h.totC;
    m = opens + commas + bare;
    n0 = opens + bare;
    if (n0 == 0 || m == 0 || m >= 0xfffffff0ull) return false;
    d_chars_.ensure(nchars + 16);
    d_str_off_.ensure(8 * (m + 1));
    d_sym_first_.ensure(8 * (n0 + 1));
    hipLaunchKernelGGL(k_tok_eds<true>, dim3(grid), dim3(256), 0, st, raw, (u64)end, a, b, c, d_chars_.as<uint8_t>(),
                       d_str_off_.as<u64>(), d_sym_first_.as<u64>(), ctl);
    const size_t pool_cap = std::max<size_t>(2 * m + 1024, 4096);
    left_.ensure(4 * pool_cap); right_.ensure(4 * pool_cap); elen_.ensure(4 * pool_cap);
    hipLaunchKernelGGL(k_tok_leaves, dim3(2048), dim3(256), 0, st, d_str_off_.as<u64>(), m, left_.as<u32>(), right_.as<u32>(),
                       elen_.as<u32>());
    for (int k = 0; k < 2; k++) { size_[k].ensure(8 * (n0 + 1)); ent_off_[k].ensure(8 * (n0 + 1)); len1_[k].ensure(8 * (n0 + 1)); }
    hipLaunchKernelGGL(k_tok_syms, dim3(2048), dim3(256), 0, st, d_sym_first_.as<u64>(), d_str_off_.as<u64>(), n0,
                       SymArrays{size_[0].as<u64>(), ent_off_[0].as<u64>(), len1_[0].as<u64>()});
    u64 sf_head[2] = {0, 0}, sf_tail[2] = {0, 0}, so_head[2] = {0, 0};
    EDSX_HIP(hipMemcpyAsync(sf_head, d_sym_first_.as<u64>(), 16, hipMemcpyDeviceToHost, st));
    EDSX_HIP(hipMemcpyAsync(sf_tail, d_sym_first_.as<u64>() + (n0 - 1), 16, hipMemcpyDeviceToHost, st));
    EDSX_HIP(hipMemcpyAsync(so_head, d_str_off_.as<u64>(), 16, hipMemcpyDeviceToHost, st));
    EDSX_HIP(hipMemcpyAsync(&h, ctl, sizeof(h), hipMemcpyDeviceToHost, st));     // `bad` of the fill pass (brace depth)
    EDSX_HIP(hipStreamSynchronize(st));
    if (h.bad) return false;
    if (linear) {
        h = TokCtl{};
        h.n = send; h.nblk = (send + TOK_BLOCK - 1) / TOK_BLOCK;
        EDSX_HIP(hipMemcpyAsync(ctl, &h, sizeof(h), hipMemcpyHostToDevice, st));
        EDSX_HIP(hipMemcpyAsync(d_raw_.ptr, seds, send, hipMemcpyHostToDevice, st));
        const unsigned sgrid = (unsigned)std::min<u64>(h.nblk, 8192);
        hipLaunchKernelGGL(k_tok_seds<false>, dim3(sgrid), dim3(256), 0, st, raw, (u64)send, a, b, c, (u64*)nullptr, 0u, ctl);
        {
            ScanSet<3> ss{{a, b, c}, {a, b, c}, {&ctl->totA, &ctl->totB, &ctl->totC}};
            exclusive_scan_multi<3>(ss, &ctl->nblk, scan_tmp_.as<u64>(), st);
        }
        EDSX_HIP(hipMemcpyAsync(&h, ctl, sizeof(h), hipMemcpyDeviceToHost, st));
        EDSX_HIP(hipStreamSynchronize(st));
        if (h.bad || (h.totA & 0xffffffffull) != m) return false;   // the host path words the error
        W = (u32)(h.maxid / 64 + 1);
        bits_.ensure(8 * pool_cap * W);
        EDSX_HIP(hipMemsetAsync(bits_.ptr, 0, 8 * (size_t)m * W, st));
        hipLaunchKernelGGL(k_tok_seds<true>, dim3(sgrid), dim3(256), 0, st, raw, (u64)send, a, b, c, bits_.as<u64>(), W, ctl);
        EDSX_HIP(hipMemcpyAsync(&h, ctl, sizeof(h), hipMemcpyDeviceToHost, st));
        EDSX_HIP(hipStreamSynchronize(st));
        if (h.bad) return false;
    }
    EDSX_HIP(hipGetLastError());
    head_single = sf_head[1] - sf_head[0] == 1;
    tail_single = sf_tail[1] - sf_tail[0] == 1;
    head_len = so_head[1] - so_head[0];
    return true;
}

// Tokenise (on the device when the text is plain, else on the host) and leave the round-0 arrays in HBM: per symbol
// size / first string / single-string length, per string its length (elen_) and, with sources, its path bitset.
void MergePipeline::prepare(const uint8_t* eds, size_t eds_n, const uint8_t* seds, size_t seds_n, bool linear, hipStream_t st,
                            Loaded& L)
{
    static const bool trace = [] { const char* e = getenv("EDSX_TRACE"); return e && atoi(e); }();
    auto t_last = std::chrono::steady_clock::now();
    auto mark = [&](const char* what) {
        if (!trace) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[edsx merge] %-26s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    // ---- tokenise: on the device when the text is plain (see "device tokenisers"), else on the host
    ctl_.ensure(8 * 16);
    u64& n0 = L.n0; u64& m = L.m; u64& head_len = L.head_len;
    u32& W = L.W;
    bool& head_single = L.head_single; bool& tail_single = L.tail_single;
    n0 = 0; m = 0; head_len = 0; W = 1; head_single = false; tail_single = false;
    const bool on_device = tokenize_device(eds, eds_n, seds, seds_n, linear, st, n0, m, W, head_single, tail_single, head_len);
    tokenised_on_device_ = on_device;
    mark(on_device ? "upload + device tokenise" : "device tokenise attempt");
    if (!on_device) {
    // ---- host tokeniser: eds.cpp:39-155 (same error texts)
    std::vector<uint8_t> chars;
    std::vector<u64> str_off{0}, sym_first{0};
    if (!tokenize_eds_parallel(eds, eds_n, chars, str_off, sym_first)) {
        chars.clear(); str_off.assign(1, 0); sym_first.assign(1, 0);
        std::string in = strip_ws(eds, eds_n);
        if (!in.empty()) {
            in = normalize(in);
            chars.reserve(in.size());
            size_t pos = 0;
            while (pos < in.size()) {
                if (in[pos] != '{') throw FormatError("Expected '{' at position " + std::to_string(pos));
                pos++;
                while (pos < in.size() && in[pos] != '}') {
                    if (in[pos] == ',') str_off.push_back(chars.size());
                    else chars.push_back((uint8_t)in[pos]);
                    pos++;
                }
                str_off.push_back(chars.size());
                if (pos >= in.size() || in[pos] != '}') throw FormatError("Expected '}' at position " + std::to_string(pos));
                pos++;
                sym_first.push_back(str_off.size() - 1);
            }
        }
    }
    n0 = sym_first.size() - 1; m = str_off.size() - 1;

    // ---- sources: eds.cpp:268-355 -> bitsets
    std::vector<u64> bits;
    std::vector<SedsPart> sparts;
    u64 par_sets = 0;
    int par_maxid = 0;
    if (linear && tokenize_seds_parallel(seds, seds_n, sparts, par_sets, par_maxid) && par_sets == m) {
        W = (u32)(par_maxid / 64 + 1);
        bits.assign((size_t)m * W, 0);
        std::vector<size_t> base(sparts.size() + 1, 0);
        for (size_t t = 0; t < sparts.size(); t++) base[t + 1] = base[t] + sparts[t].set_end.size();
        std::vector<std::thread> th;
        for (size_t t = 0; t < sparts.size(); t++)
            th.emplace_back([&, t] {
                const SedsPart& pt = sparts[t];
                size_t b = 0;
                for (size_t sidx = 0; sidx < pt.set_end.size(); sidx++) {
                    for (; b < pt.set_end[sidx]; b++) { const int id = pt.ids[b]; bits[(base[t] + sidx) * W + id / 64] |= 1ull << (id % 64); }
                }
            });
        for (auto& x : th) x.join();
    } else if (linear) {
        std::string in = strip_ws(seds, seds_n);
        if (in.empty()) throw FormatError("sEDS input is empty");
        std::vector<std::vector<int>> sets;
        int maxid = 0;
        size_t pos = 0;
        while (pos < in.size()) {
            if (in[pos] != '{') throw FormatError("sEDS: Expected '{' at position " + std::to_string(pos));
            pos++;
            std::vector<int> ids;
            std::string num;
            auto flush = [&] {
                if (num.empty()) return;
                int id;
                try { id = std::stoi(num); } catch (...) { throw FormatError("stoi"); }
                ids.push_back(id);
                maxid = std::max(maxid, id);
                num.clear();
            };
            while (pos < in.size() && in[pos] != '}') {
                if (in[pos] == ',') flush();
                else if (std::isdigit((unsigned char)in[pos])) num += in[pos];
                else
                    throw FormatError("sEDS: Invalid character '" + std::string(1, in[pos]) + "' at position " +
                                      std::to_string(pos));
                pos++;
            }
            flush();
            if (pos >= in.size() || in[pos] != '}') throw FormatError("sEDS: Expected '}' at position " + std::to_string(pos));
            pos++;
            if (ids.empty()) throw FormatError("sEDS: Empty path set at string " + std::to_string(sets.size()));
            sets.push_back(std::move(ids));
        }
        if (sets.size() != m)
            throw FormatError("sEDS: Source count (" + std::to_string(sets.size()) + ") does not match EDS cardinality (" +
                              std::to_string(m) + ")");
        W = (u32)(maxid / 64 + 1);
        bits.assign((size_t)m * W, 0);
        for (size_t sidx = 0; sidx < m; sidx++)
            for (int id : sets[sidx]) bits[sidx * W + id / 64] |= 1ull << (id % 64);
    }

    if (n0 == 0) { L.n0 = 0; L.m = 0; L.W = 1; return; }      // empty EDS
    if (m >= 0xfffffff0ull) throw FormatError("EDS has too many strings for this build");
    head_single = sym_first[1] - sym_first[0] == 1; tail_single = sym_first[n0] - sym_first[n0 - 1] == 1;
    head_len = str_off[1] - str_off[0];

    // ---- upload
    d_chars_.ensure(chars.size() + 16);
    d_str_off_.ensure(8 * (m + 1));
    EDSX_HIP(hipMemcpyAsync(d_chars_.ptr, chars.data(), chars.size(), hipMemcpyHostToDevice, st));
    EDSX_HIP(hipMemcpyAsync(d_str_off_.ptr, str_off.data(), 8 * (m + 1), hipMemcpyHostToDevice, st));
    size_t pool_cap = std::max<size_t>(2 * m + 1024, 4096);
    left_.ensure(4 * pool_cap); right_.ensure(4 * pool_cap); elen_.ensure(4 * pool_cap);
    if (linear) bits_.ensure(8 * pool_cap * W);
    {
        std::vector<u32> hl(m, LEAF), hr(m), he(m);
        for (u64 s = 0; s < m; s++) { hr[s] = (u32)s; he[s] = (u32)(str_off[s + 1] - str_off[s]); }
        EDSX_HIP(hipMemcpyAsync(left_.ptr, hl.data(), 4 * m, hipMemcpyHostToDevice, st));
        EDSX_HIP(hipMemcpyAsync(right_.ptr, hr.data(), 4 * m, hipMemcpyHostToDevice, st));
        EDSX_HIP(hipMemcpyAsync(elen_.ptr, he.data(), 4 * m, hipMemcpyHostToDevice, st));
        if (linear) EDSX_HIP(hipMemcpyAsync(bits_.ptr, bits.data(), 8 * (size_t)m * W, hipMemcpyHostToDevice, st));
        EDSX_HIP(hipStreamSynchronize(st));
    }
    for (int b = 0; b < 2; b++) { size_[b].ensure(8 * (n0 + 1)); ent_off_[b].ensure(8 * (n0 + 1)); len1_[b].ensure(8 * (n0 + 1)); }
    {
        std::vector<u64> hs(n0), ho(n0), hl(n0);
        for (u64 i = 0; i < n0; i++) {
            hs[i] = sym_first[i + 1] - sym_first[i];
            ho[i] = sym_first[i];
            hl[i] = hs[i] == 1 ? str_off[sym_first[i] + 1] - str_off[sym_first[i]] : 0;
        }
        EDSX_HIP(hipMemcpyAsync(size_[0].ptr, hs.data(), 8 * n0, hipMemcpyHostToDevice, st));
        EDSX_HIP(hipMemcpyAsync(ent_off_[0].ptr, ho.data(), 8 * n0, hipMemcpyHostToDevice, st));
        EDSX_HIP(hipMemcpyAsync(len1_[0].ptr, hl.data(), 8 * n0, hipMemcpyHostToDevice, st));
        EDSX_HIP(hipStreamSynchronize(st));
    }
    }
    if (!on_device) mark("host tokenise + upload");
}

// ---- statistics and l-EDS validity as device reductions (eds.cpp:361-505, eds_transforms.cpp:439-468) -------------
// acc: [0] degenerate symbols [1] sum(size - 1) over them [2] characters of the non-degenerate symbols [3] their number
//      [4] min / [5] max of their lengths [6] "not an l-EDS" [7] empty strings [8] all characters [9] sum of the set
//      sizes [10] largest set; orbits[W]: OR of all path sets
__device__ __forceinline__ u64 wave_sum64(u64 v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__global__ void __launch_bounds__(256) k_eds_stats(SymArrays s, u64 n, const u32* __restrict__ elen, u64 m, const u64* __restrict__ bits,
                                                   u32 W, u64 l, u64* __restrict__ acc, u64* __restrict__ orbits)
{
    const u64 t0 = blockIdx.x * (u64)blockDim.x + threadIdx.x, step = (u64)gridDim.x * blockDim.x;
    u64 ndeg = 0, change = 0, common = 0, nctx = 0, mn = ~0ull, mx = 0, bad = 0;
    for (u64 i = t0; i < n; i += step) {
        const u64 sz = s.size[i];
        if (sz > 1) {
            ndeg++; change += sz - 1;
            if (i + 1 < n && s.size[i + 1] > 1) bad = 1;        // adjacent degenerate symbols (:462-464)
        } else {
            const u64 len = s.len1[i];
            common += len; nctx++;
            mn = len < mn ? len : mn; mx = len > mx ? len : mx;
            if (l && i > 0 && i + 1 < n && len < l) bad = 1;     // a short internal common block (:455-457)
        }
    }
    u64 empty = 0, chars = 0, tot = 0, big = 0;
    for (u64 k = t0; k < m; k += step) {
        const u64 e = elen[k];
        empty += e == 0; chars += e;
        if (bits) {
            u64 c = 0;
            for (u32 w = 0; w < W; w++) c += (u64)__builtin_popcountll(bits[k * W + w]);
            tot += c; big = c > big ? c : big;
        }
    }
    if (bits) {                                                   // OR of all sets, word by word
        for (u32 w = 0; w < W; w++) {
            u64 o = 0;
            for (u64 k = t0; k < m; k += step) o |= bits[k * W + w];
            for (int sh = 32; sh > 0; sh >>= 1) o |= __shfl_xor(o, sh, 64);
            if ((threadIdx.x & 63) == 0 && o) atomicOr(&orbits[w], o);
        }
    }
    ndeg = wave_sum64(ndeg); change = wave_sum64(change); common = wave_sum64(common); nctx = wave_sum64(nctx);
    empty = wave_sum64(empty); chars = wave_sum64(chars); tot = wave_sum64(tot);
    for (int sh = 32; sh > 0; sh >>= 1) {
        const u64 a = __shfl_xor(mn, sh, 64), b = __shfl_xor(mx, sh, 64), c = __shfl_xor(big, sh, 64), d = __shfl_xor(bad, sh, 64);
        mn = a < mn ? a : mn; mx = b > mx ? b : mx; big = c > big ? c : big; bad |= d;
    }
    if ((threadIdx.x & 63) == 0) {
        if (ndeg) atomicAdd(&acc[0], ndeg);
        if (change) atomicAdd(&acc[1], change);
        if (common) atomicAdd(&acc[2], common);
        if (nctx) atomicAdd(&acc[3], nctx);
        atomicMin(&acc[4], mn); atomicMax(&acc[5], mx);
        if (bad) atomicOr(&acc[6], 1ull);
        if (empty) atomicAdd(&acc[7], empty);
        if (chars) atomicAdd(&acc[8], chars);
        if (tot) atomicAdd(&acc[9], tot);
        atomicMax(&acc[10], big);
    }
}

void MergePipeline::stats(const uint8_t* eds, size_t eds_n, const uint8_t* seds, size_t seds_n, uint32_t l, EdsStats& out,
                          hipStream_t st)
{
    const bool linear = seds != nullptr;
    Loaded L;
    prepare(eds, eds_n, seds, seds_n, linear, st, L);
    out = EdsStats{};
    out.has_sources = linear ? 1 : 0;
    out.is_leds = 1;
    if (L.n0 == 0) return;                                      // empty EDS: all zero (eds.cpp:362-376), trivially an l-EDS
    const u32 W = linear ? L.W : 0;
    a_.ensure(8 * (16 + (size_t)W + 1));
    u64* acc = a_.as<u64>();
    std::vector<u64> h(16 + W, 0);
    h[4] = ~0ull;
    EDSX_HIP(hipMemcpyAsync(acc, h.data(), 8 * h.size(), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_eds_stats, dim3(1024), dim3(256), 0, st, SymArrays{size_[0].as<u64>(), ent_off_[0].as<u64>(), len1_[0].as<u64>()},
                       L.n0, elen_.as<u32>(), L.m, linear ? bits_.as<u64>() : nullptr, W, (u64)l, acc, acc + 16);
    EDSX_HIP(hipMemcpyAsync(h.data(), acc, 8 * h.size(), hipMemcpyDeviceToHost, st));
    EDSX_HIP(hipStreamSynchronize(st));
    EDSX_HIP(hipGetLastError());
    out.n_symbols = L.n0; out.n_strings = L.m; out.n_chars = h[8];
    out.num_degenerate = h[0]; out.total_change_size = h[1]; out.num_common_chars = h[2]; out.num_context_blocks = h[3];
    out.min_context = h[3] ? h[4] : 0; out.max_context = h[5]; out.num_empty_strings = h[7];
    out.is_leds = (l == 0 || !h[6]) ? 1 : 0;
    if (linear) {
        out.total_paths = h[9]; out.max_paths_per_string = h[10];
        for (u32 w = 0; w < W; w++) out.num_paths += (u64)__builtin_popcountll(h[16 + w]);
    }
}

void MergePipeline::run(const uint8_t* eds, size_t eds_n, const uint8_t* seds, size_t seds_n, uint32_t l, bool compact,
                        HostBytes& out, HostBytes& seds_out, hipStream_t st, MergeShard* shard)
{
    if (l == 0) throw ParamError("context_length must be > 0 for l-EDS transformation");   // :322-324
    const bool linear = seds != nullptr;
    // EDSX_TRACE=1: wall-clock of the host-visible stages on stderr (every mark follows a stream synchronisation)
    static const bool trace = [] { const char* e = getenv("EDSX_TRACE"); return e && atoi(e); }();
    auto t_last = std::chrono::steady_clock::now();
    auto mark = [&](const char* what) {
        if (!trace) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[edsx merge] %-26s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };

    Loaded L;
    prepare(eds, eds_n, seds, seds_n, linear, st, L);
    t_last = std::chrono::steady_clock::now();
    const u64 n0 = L.n0, m = L.m, head_len = L.head_len;
    const u32 W = L.W;
    const bool head_single = L.head_single, tail_single = L.tail_single;
    if (n0 == 0) {                                           // empty EDS: save() prints "\n" (quirk 19)
        out.take(1);
        out.data[0] = '\n';
        seds_out.take(0);
        return;
    }
    if (shard) {
        shard->head_intact = shard->tail_intact = true;
        if ((shard->head_sentinel && !head_single) || (shard->tail_sentinel && !tail_single) ||
            ((shard->head_sentinel && shard->tail_sentinel) && n0 < 3))
            throw ParamError("a sentinel of a symbol range must be a single-string symbol of its own");
    }
    a_.ensure(8 * (n0 + 2)); b_.ensure(8 * (n0 + 2)); c_.ensure(8 * (n0 + 2)); d_.ensure(8 * (n0 + 2)); e_.ensure(8 * (n0 + 2));
    scan_tmp_.ensure(8 * ((n0 + 2) / SCAN_TILE + 2));
    u64* ctl = ctl_.as<u64>();         // [0]=n  [1]=any  [2]=n_new  [3]=total_new  [4]=err_pos  [5]=err_big  [6]=scratch

    u64 n = n0, pool_used = m;
    int cur = 0;
    size_t iteration = 0;
    const size_t MAX_ITERATIONS = 10000;                     // eds_transforms.cpp:335
    while (iteration < MAX_ITERATIONS) {
        if (n < 2) break;
        u64 hctl[8] = {n, 0, 0, 0, NO_ERR, NO_ERR, 0, 0};
        EDSX_HIP(hipMemcpyAsync(ctl, hctl, sizeof(hctl), hipMemcpyHostToDevice, st));
        SymArrays sc{size_[cur].as<u64>(), ent_off_[cur].as<u64>(), len1_[cur].as<u64>()};
        SymArrays sn{size_[cur ^ 1].as<u64>(), ent_off_[cur ^ 1].as<u64>(), len1_[cur ^ 1].as<u64>()};
        Pool pool{left_.as<u32>(), right_.as<u32>(), elen_.as<u32>(), bits_.as<u64>(), W};
        u64 *runmark = a_.as<u64>(), *sel = b_.as<u64>(), *keep = c_.as<u64>(), *newidx = d_.as<u64>(), *newcnt = e_.as<u64>();
        hipLaunchKernelGGL(k_should, dim3(1024), dim3(256), 0, st, sc, n, (u64)l, runmark, ctl + 1);
        inclusive_max_scan_u64(runmark, runmark, ctl + 0, ctl + 6, scan_tmp_.as<u64>(), st);
        hipLaunchKernelGGL(k_select, dim3(1024), dim3(256), 0, st, runmark, n, sel, keep);
        hipLaunchKernelGGL(k_keep, dim3(1024), dim3(256), 0, st, sel, n, keep);
        exclusive_scan_u64(keep, newidx, ctl + 0, ctl + 2, scan_tmp_.as<u64>(), st);
        RoundParams rp{sc, sn, pool, n, sel, keep, newidx, newcnt, newcnt, pool_used, ctl + 4, ctl + 5, linear ? 1 : 0};
        hipLaunchKernelGGL(k_pair_count, dim3(1024), dim3(256), 0, st, rp);
        exclusive_scan_u64(newcnt, newcnt, ctl + 2, ctl + 3, scan_tmp_.as<u64>(), st);
        EDSX_HIP(hipMemcpyAsync(hctl, ctl, sizeof(hctl), hipMemcpyDeviceToHost, st));
        EDSX_HIP(hipStreamSynchronize(st));
        if (!hctl[1]) break;                                 // is_leds: nothing left to merge
        if (hctl[4] != NO_ERR)
            throw FormatError("Merging positions " + std::to_string(hctl[4]) + " and " + std::to_string(hctl[4] + 1) +
                              " results in empty set (no valid source intersections)");
        if (hctl[5] != NO_ERR || pool_used + hctl[3] >= 0xfffffff0ull)
            throw FormatError("l-EDS merge produces more than 2^32 strings; refusing (the reference would exhaust memory)");
        const u64 n_new = hctl[2], total_new = hctl[3];
        const size_t need = pool_used + total_new + 16;
        grow_keep(left_, 4 * pool_used, 4 * need, st);
        grow_keep(right_, 4 * pool_used, 4 * need, st);
        grow_keep(elen_, 4 * pool_used, 4 * need, st);
        if (linear) grow_keep(bits_, 8 * pool_used * W, 8 * need * W, st);
        rp.pool = Pool{left_.as<u32>(), right_.as<u32>(), elen_.as<u32>(), bits_.as<u64>(), W};
        hipLaunchKernelGGL(k_pair_fill, dim3(1024), dim3(256), 0, st, rp);
        pool_used += total_new;
        n = n_new;
        cur ^= 1;
        iteration++;
    }
    if (iteration >= MAX_ITERATIONS) throw FormatError("Maximum iterations reached without convergence");
    rounds_run_ = iteration;
    mark("merge rounds");

    // ---- symbol range of a partitioned merge: did the sentinels stay out of every merge?  Merged symbols get
    // fresh pool entries (>= m), an untouched sentinel still points at its leaf.
    if (shard && (shard->head_sentinel || shard->tail_sentinel)) {
        u64 h[4] = {0, 0, 0, 0};
        EDSX_HIP(hipMemcpyAsync(h + 0, size_[cur].as<u64>(), 8, hipMemcpyDeviceToHost, st));
        EDSX_HIP(hipMemcpyAsync(h + 1, ent_off_[cur].as<u64>(), 8, hipMemcpyDeviceToHost, st));
        EDSX_HIP(hipMemcpyAsync(h + 2, size_[cur].as<u64>() + (n - 1), 8, hipMemcpyDeviceToHost, st));
        EDSX_HIP(hipMemcpyAsync(h + 3, ent_off_[cur].as<u64>() + (n - 1), 8, hipMemcpyDeviceToHost, st));
        EDSX_HIP(hipStreamSynchronize(st));
        if (shard->head_sentinel) shard->head_intact = h[0] == 1 && h[1] == 0 && n >= 2;
        if (shard->tail_sentinel) shard->tail_intact = h[2] == 1 && h[3] == m - 1 && n >= 2;
    }

    // ---- final text
    SymArrays sf{size_[cur].as<u64>(), ent_off_[cur].as<u64>(), len1_[cur].as<u64>()};
    Pool pool{left_.as<u32>(), right_.as<u32>(), elen_.as<u32>(), bits_.as<u64>(), W};
    u64 hctl[8] = {n, 0, 0, 0, 0, 0, 0, 0};
    EDSX_HIP(hipMemcpyAsync(ctl, hctl, sizeof(hctl), hipMemcpyHostToDevice, st));
    u64* cum = a_.as<u64>();
    exclusive_scan_u64(sf.size, cum, ctl + 0, ctl + 1, scan_tmp_.as<u64>(), st);
    EDSX_HIP(hipMemcpyAsync(hctl, ctl, sizeof(hctl), hipMemcpyDeviceToHost, st));
    EDSX_HIP(hipStreamSynchronize(st));
    const u64 nstr = hctl[1];
    if (nstr >= 0xfffffff0ull) throw FormatError("l-EDS output has too many strings");
    fin_ent_.ensure(4 * (nstr + 1)); fin_flag_.ensure(nstr + 16); fbytes_.ensure(8 * (nstr + 2)); fsbytes_.ensure(8 * (nstr + 2));
    scan_tmp_.ensure(8 * ((nstr + 2) / SCAN_TILE + 2));
    FinParams fp{sf, pool, n, cum, fin_ent_.as<u32>(), fin_flag_.as<uint8_t>(), fbytes_.as<u64>(), fsbytes_.as<u64>(),
                 d_chars_.as<uint8_t>(), d_str_off_.as<u64>(), nullptr, nullptr, ctl + 4, compact ? 1 : 0, linear ? 1 : 0};
    hipLaunchKernelGGL(k_fin_list, dim3(1024), dim3(256), 0, st, fp);
    hctl[0] = nstr; hctl[4] = 0;
    EDSX_HIP(hipMemcpyAsync(ctl, hctl, sizeof(hctl), hipMemcpyHostToDevice, st));
    exclusive_scan_u64(fbytes_.as<u64>(), fbytes_.as<u64>(), ctl + 0, ctl + 2, scan_tmp_.as<u64>(), st);
    if (linear) exclusive_scan_u64(fsbytes_.as<u64>(), fsbytes_.as<u64>(), ctl + 0, ctl + 3, scan_tmp_.as<u64>(), st);
    EDSX_HIP(hipMemcpyAsync(hctl, ctl, sizeof(hctl), hipMemcpyDeviceToHost, st));
    EDSX_HIP(hipStreamSynchronize(st));
    const u64 E = hctl[2], Q = linear ? hctl[3] : 0;
    d_out_.ensure(E + 16);
    if (linear) d_sout_.ensure(Q + 16);
    fp.out = d_out_.as<uint8_t>(); fp.sout = d_sout_.as<uint8_t>();
    if (rounds_run_ + 2 > (u64)FIN_STACK && nstr) {          // trees may be deeper than the serial walk's own stack
        fin_spill_.ensure(4 * nstr * (rounds_run_ + 2));
        fp.spill = fin_spill_.as<u32>(); fp.spill_depth = (u32)(rounds_run_ + 2);
    }
    {
        // strings per workgroup: about 16 KB of text (a power of two, at most one string per thread)
        const u64 mean = nstr ? std::max<u64>(1, E / nstr) : 1;
        u32 ns = 256;
        while (ns > 1 && (u64)ns * mean > 16384) ns >>= 1;
        if (nstr) hipLaunchKernelGGL(k_fin_write, dim3((unsigned)std::min<u64>((nstr + ns - 1) / ns, 1u << 20)), dim3(256), 0, st, fp, nstr, E, ns);
        if (linear && nstr) hipLaunchKernelGGL(k_fin_write_sources, dim3(2048), dim3(256), 0, st, fp, nstr);
    }
    mark("final sizes");
    // malloc'ed, not value-initialised (see HostBytes): the pages are first touched by the copy
    out.take(E + 1);
    PinnedDownload::copy(out.data, d_out_.ptr, E, st);
    if (linear) {
        seds_out.take(Q + 1);
        PinnedDownload::copy(seds_out.data, d_sout_.ptr, Q, st);
    } else seds_out.take(0);
    EDSX_HIP(hipMemcpyAsync(hctl, ctl, sizeof(hctl), hipMemcpyDeviceToHost, st));
    EDSX_HIP(hipStreamSynchronize(st));
    EDSX_HIP(hipGetLastError());
    if (hctl[4]) throw FormatError("l-EDS merge nesting deeper than this build supports");
    mark("text kernel + download");
    out.data[E] = '\n';                                      // eds.cpp:630
    if (linear) seds_out.data[Q] = '\n';                     // eds.cpp:658
    if (shard && shard->tail_sentinel) {                     // not the last range: the text goes on
        out.size--;
        if (linear) seds_out.size--;
    }
    if (shard && shard->head_sentinel && shard->head_intact) {   // the left neighbour prints the shared sentinel
        out.drop_front((size_t)head_len + (compact ? 0 : 2));
        if (linear) {
            const void* b = memchr(seds_out.data, '}', seds_out.size);
            seds_out.drop_front(b ? static_cast<size_t>(static_cast<const uint8_t*>(b) - seds_out.data) + 1 : 0);
        }
    }
}

} // namespace edsx
