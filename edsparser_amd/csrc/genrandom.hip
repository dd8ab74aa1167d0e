// genrandom.hip — random EDS with controlled variability + its sources, generated directly in HBM.
//
// Shape and flags of the reference generator (src/cpp/tools/genrandomeds.cpp:221-352 text, :64-112 site placement,
// :121-149 path choices, :158-186 source sets): a reference of total_bp characters uniform over the alphabet; variant
// sites are single positions; a site has k ~ U[min_alt, max_alt] alternatives, the first is the reference character,
// each other one is a SNP to a different character with probability snp_ratio, else an insertion (reference character +
// U[1, var_len_max] random characters) or a deletion (empty string) with equal odds; P = max(max_alt, 3) paths, path
// a < k takes alternative a, the others choose uniformly; a degenerate string's source set lists the 1-based paths that
// chose it, a common block has {0}; FULL brackets, no trailing newline.
// Not the reference's byte stream: it draws from std::mt19937 sequentially, this generator is counter-based (every
// decision is a hash of (seed, position, purpose)), so any part of the text can be produced independently on any GPU.
// Site placement: min_context == 0 -> every position is a site with probability `variability` (the reference draws
// exactly floor(total_bp * variability) distinct positions: the count here is binomial around that); min_context > 0 ->
// the reference's own rule, one site per segment of total_bp / n_sites positions at base + U[0, min(segment - 1, min_context)].
//
// Three passes over the positions: text bytes of every position (.eds / .seds), two exclusive scans, text.
#include "genrandom.hpp"

namespace edsx {

namespace {

struct GenDev {
    u64 total_bp, thr /* 2^40 * variability */, min_context, seg_size, n_seg, seed;
    u32 min_alt, max_alt, var_len_max, alpha_n, snp_thr /* 2^24 * snp_ratio */, n_paths;
    uint8_t alphabet[64];
};

__device__ __forceinline__ u32 ref_char(const GenDev& g, u64 p) { return g.alphabet[hash3(g.seed, p, 3) % g.alpha_n]; }

__device__ __forceinline__ bool is_site(const GenDev& g, u64 p)
{
    if (g.min_context == 0) return (hash3(g.seed, p, 1) >> 24) < g.thr;
    if (g.n_seg == 0 || g.seg_size == 0) return false;
    // genrandomeds.cpp:95-108: site i at min(i * segment + min_context + U[0, min(segment - 1, min_context)], total_bp - 1)
    const u64 lim = g.seg_size - 1 < g.min_context ? g.seg_size - 1 : g.min_context;
    u64 i = p >= g.min_context ? (p - g.min_context) / g.seg_size : 0;
    for (u64 k = i > 0 ? i - 1 : 0; k <= i && k < g.n_seg; k++) {           // the offset may reach into the next segment
        const u64 base = k * g.seg_size + g.min_context;
        if (base >= g.total_bp) continue;
        u64 pos = base + hash3(g.seed, k, 5) % (lim + 1);
        if (pos > g.total_bp - 1) pos = g.total_bp - 1;
        if (pos == p) return true;
    }
    return false;
}

struct Site { u32 k; u32 kind[16]; u32 ins[16]; };                         // kind: 0 reference, 1 SNP, 2 insertion, 3 deletion
__device__ __forceinline__ Site site_of(const GenDev& g, u64 p)
{
    Site s;
    s.k = g.min_alt + (u32)(hash3(g.seed, p, 2) % (g.max_alt - g.min_alt + 1));
    s.kind[0] = 0; s.ins[0] = 0;
    for (u32 a = 1; a < s.k; a++) {
        const u64 t = hash3(g.seed, p, 16 + a);
        s.ins[a] = 0;
        if ((u32)(t & 0xffffffu) < g.snp_thr) s.kind[a] = 1;
        else if ((t >> 24) & 1) { s.kind[a] = 2; s.ins[a] = 1 + (u32)((t >> 32) % g.var_len_max); }
        else s.kind[a] = 3;
    }
    return s;
}
__device__ __forceinline__ u32 snp_char(const GenDev& g, u64 p, u32 a, u32 refc)
{   // a character of the alphabet different from the reference one (the reference character itself if there is no other)
    u32 others = 0;
    for (u32 i = 0; i < g.alpha_n; i++) others += g.alphabet[i] != refc;
    if (!others) return refc;
    u32 pick = (u32)(hash3(g.seed, p, 32 + a) % others);
    for (u32 i = 0; i < g.alpha_n; i++)
        if (g.alphabet[i] != refc) { if (!pick) return g.alphabet[i]; pick--; }
    return refc;
}
__device__ __forceinline__ u32 path_choice(const GenDev& g, u64 p, u32 path, u32 k)
{
    return path < k ? path : (u32)(hash3(g.seed ^ 0x5eed5eedull, p, 256 + path) % k);      // genrandomeds.cpp:134-146
}
__device__ __forceinline__ u32 ndig(u32 v) { return v >= 100 ? 3 : v >= 10 ? 2 : 1; }

template <bool EMIT>
__global__ void __launch_bounds__(256) k_gen(GenDev g, u64* __restrict__ eb, u64* __restrict__ sb, uint8_t* __restrict__ eds,
                                             uint8_t* __restrict__ seds, u64* __restrict__ n_sites)
{
    u64 mine = 0;
    for (u64 p = blockIdx.x * (u64)blockDim.x + threadIdx.x; p < g.total_bp; p += (u64)gridDim.x * blockDim.x) {
        const bool site = is_site(g, p);
        if (!site) {
            const bool first = p == 0 || is_site(g, p - 1), last = p + 1 == g.total_bp || is_site(g, p + 1);
            if (!EMIT) { eb[p] = 1 + (first ? 1 : 0) + (last ? 1 : 0); sb[p] = first ? 3 : 0; }
            else {
                uint8_t* e = eds + eb[p];
                if (first) { *e++ = '{'; uint8_t* s = seds + sb[p]; s[0] = '{'; s[1] = '0'; s[2] = '}'; }
                *e++ = (uint8_t)ref_char(g, p);
                if (last) *e = '}';
            }
            continue;
        }
        mine++;
        const Site s = site_of(g, p);
        const u32 refc = ref_char(g, p);
        if (!EMIT) {
            u64 e = 1 + s.k;                                      // '{', separators and '}'
            for (u32 a = 0; a < s.k; a++) e += s.kind[a] == 3 ? 0 : 1 + s.ins[a];
            u64 q = 0;
            for (u32 a = 0; a < s.k; a++) {
                u32 cnt = 0, dig = 0;
                for (u32 path = 0; path < g.n_paths; path++)
                    if (path_choice(g, p, path, s.k) == a) { cnt++; dig += ndig(path + 1); }
                q += 2 + dig + (cnt ? cnt - 1 : 0);              // '{' ids ',' .. '}'  (cnt >= 1: path a takes alternative a)
            }
            eb[p] = e; sb[p] = q;
        } else {
            uint8_t* e = eds + eb[p];
            *e++ = '{';
            for (u32 a = 0; a < s.k; a++) {
                if (a) *e++ = ',';
                if (s.kind[a] == 0) *e++ = (uint8_t)refc;
                else if (s.kind[a] == 1) *e++ = (uint8_t)snp_char(g, p, a, refc);
                else if (s.kind[a] == 2) {
                    *e++ = (uint8_t)refc;
                    for (u32 i = 0; i < s.ins[a]; i++) *e++ = g.alphabet[hash3(g.seed, p, 64 + a * 64 + i) % g.alpha_n];
                }
            }
            *e = '}';
            uint8_t* q = seds + sb[p];
            for (u32 a = 0; a < s.k; a++) {
                *q++ = '{';
                bool any = false;
                for (u32 path = 0; path < g.n_paths; path++) {
                    if (path_choice(g, p, path, s.k) != a) continue;
                    if (any) *q++ = ',';
                    const u32 id = path + 1;
                    if (id >= 100) *q++ = (uint8_t)('0' + id / 100);
                    if (id >= 10) *q++ = (uint8_t)('0' + (id / 10) % 10);
                    *q++ = (uint8_t)('0' + id % 10);
                    any = true;
                }
                *q++ = '}';
            }
        }
    }
    if (!EMIT && mine) atomicAdd(n_sites, mine);
}

} // namespace

void GenPipeline::run(const GenParams& p, HostBytes& eds, HostBytes& seds, u64& n_sites, hipStream_t st)
{
    if (p.total_bp == 0) throw ParamError("Reference size must be greater than 0");
    if (!(p.variability >= 0.0 && p.variability <= 1.0)) throw ParamError("Variability must be between 0.0 and 1.0");   // (NaN fails too)
    if (p.min_alt < 2) throw ParamError("Minimum alternatives must be at least 2");
    if (p.max_alt < p.min_alt) throw ParamError("Maximum alternatives must be >= minimum alternatives");
    if (p.max_alt > 16) throw ParamError("Maximum alternatives above 16 are not supported by this build");
    if (p.var_len_max == 0) throw ParamError("Variant length max must be greater than 0");
    if (p.var_len_max > 63) throw ParamError("Variant length max above 63 is not supported by this build");
    if (!(p.snp_ratio >= 0.0 && p.snp_ratio <= 1.0)) throw ParamError("SNP ratio must be between 0.0 and 1.0");
    if (p.alpha_n == 0 || p.alpha_n > 64) throw ParamError("Alphabet cannot be empty");
    GenDev g{};
    g.total_bp = p.total_bp; g.thr = (u64)(p.variability * 1099511627776.0); g.min_context = p.min_context; g.seed = p.seed;
    g.min_alt = p.min_alt; g.max_alt = p.max_alt; g.var_len_max = p.var_len_max; g.alpha_n = p.alpha_n;
    g.snp_thr = (u32)(p.snp_ratio * 16777216.0); g.n_paths = std::max<u32>(p.max_alt, 3);      // genrandomeds.cpp:244
    std::memcpy(g.alphabet, p.alphabet, p.alpha_n);
    if (p.min_context) {                                         // genrandomeds.cpp:86-98
        const u64 want = (u64)((double)p.total_bp * p.variability), fit = p.total_bp / (p.min_context + 1);
        g.n_seg = std::min(want, fit);
        g.seg_size = g.n_seg ? p.total_bp / g.n_seg : 0;
    }
    const u64 n = p.total_bp;
    ebytes_.ensure(8 * (n + 2)); sbytes_.ensure(8 * (n + 2)); scan_tmp_.ensure(8 * ((n + 2) / SCAN_TILE + 4)); ctl_.ensure(64);
    u64* ctl = ctl_.as<u64>();                                   // [0] n  [1] E  [2] Q  [3] sites
    u64 h[4] = {n, 0, 0, 0};
    EDSX_HIP(hipMemcpyAsync(ctl, h, sizeof(h), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_gen<false>, dim3(4096), dim3(256), 0, st, g, ebytes_.as<u64>(), sbytes_.as<u64>(), nullptr, nullptr, ctl + 3);
    exclusive_scan_u64(ebytes_.as<u64>(), ebytes_.as<u64>(), ctl + 0, ctl + 1, scan_tmp_.as<u64>(), st);
    exclusive_scan_u64(sbytes_.as<u64>(), sbytes_.as<u64>(), ctl + 0, ctl + 2, scan_tmp_.as<u64>(), st);
    EDSX_HIP(hipMemcpyAsync(h, ctl, sizeof(h), hipMemcpyDeviceToHost, st));
    EDSX_HIP(hipStreamSynchronize(st));
    const u64 E = h[1], Q = h[2];
    n_sites = h[3];
    out_e_.ensure(E + 16); out_s_.ensure(Q + 16);
    hipLaunchKernelGGL(k_gen<true>, dim3(4096), dim3(256), 0, st, g, ebytes_.as<u64>(), sbytes_.as<u64>(), out_e_.as<uint8_t>(),
                       out_s_.as<uint8_t>(), ctl + 3);
    eds.take(E); seds.take(Q);
    PinnedDownload::copy(eds.data, out_e_.ptr, E, st);
    PinnedDownload::copy(seds.data, out_s_.ptr, Q, st);
    EDSX_HIP(hipGetLastError());
}

} // namespace edsx
