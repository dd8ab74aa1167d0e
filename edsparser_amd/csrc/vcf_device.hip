// vcf_device.hip — VCF (+ reference FASTA) -> EDS / sEDS on gfx950: the variant-overlay walk.
//
// Host (this file, C++): FASTA metadata, std::sort by POS, and the VCF tokeniser for files the device
// tokeniser does not accept — the text-parsing front end of src/cpp/lib/transforms/vcf_transforms.cpp
// (:51-86, :142-326, :715-718).  Plain files are tokenised on the device (k_vt_*).
// Device (HIP kernels): everything from the sorted records on —
//   group_overlapping_variants :482-534   -> max-scan of the record ends + k_mark_groups
//   read_fasta_region :98-129             -> k_fa_count/k_fa_compact (newline-free reference stream)
//   merge_variant_group :396-476          -> k_grp_count / k_grp_haps / k_grp_samples
//   generate_eds_from_variants :554-668   -> k_grp_sizes / k_grp_emit
//
// Data layout in HBM
//   refc            the FASTA bytes from the first sequence line to EOF with '\n' and '\r' removed
//                   (exactly the characters read_fasta_region can return), blkpre[] = kept bytes
//                   before every 256-byte block, so any file offset maps to a position in refc.
//   records (SoA)   start (0-based), reflen, alt range -> alt strings, per (record, sample) allele
//                   lists (two-level CSR), all in sorted order.
//   groups          first record, span [gs, gs+spanlen) in refc, raw haplotypes materialised once in
//                   hapchars, canonical (deduplicated) index per raw haplotype, carried[hap][sample]
//                   bit matrix.
#include "vcf_device.hpp"


#include <algorithm>
#include <cstring>
#include <sstream>
#include <chrono>
#include <thread>

namespace edsx {

struct VcfDev {
    // reference
    const uint8_t* fasta; u64 fasta_n; u64 seq_start, seq_size, lw;
    const uint8_t* refc; u64 refc_n; const u64* blkpre;
    // records
    u64 nrec; const u64* start; const u64* reflen; const u64* alt0; const u64* altstr_off; const uint8_t* altchars;
    const u64* pair0;          // per record: first (record,sample) pair; pair0[nrec] = #pairs
    const u64* pair_a0;        // per pair: first allele; pair_a0[#pairs] = #alleles
    const int* alleles;
    // groups
    u64 ngrp; const u64* grp_r0;   // first record of group g, grp_r0[ngrp] = nrec
    const u64* incl_end;           // inclusive max-scan of record ends
    u64 cur0;                      // reference position the walk starts at (0 unless this is a later position range)
};

__device__ __forceinline__ u64 fa_cpos(const VcfDev& d, u64 file_off)
{   // position in refc of file offset file_off (>= seq_start); offsets past EOF map to refc_n
    if (file_off >= d.fasta_n) return d.refc_n;
    const u64 rel = file_off - d.seq_start;
    u64 c = d.blkpre[rel >> 8];
    // sequence bytes in [block start, file_off): eight bytes per (unaligned) load, line breaks counted with an exact
    // SWAR zero-byte test; the buffer has 16 bytes of slack behind fasta_n, bytes at or past file_off are masked out
    u64 f = d.seq_start + (rel & ~255ull);
    u64 breaks = 0;
    while (f < file_off) {
        u64 w;
        __builtin_memcpy(&w, d.fasta + f, 8);
        const u64 left = file_off - f;
        const u64 keep = left >= 8 ? ~0ull : ((1ull << (8 * left)) - 1ull);
        const u64 a = w ^ 0x0a0a0a0a0a0a0a0aull, b = w ^ 0x0d0d0d0d0d0d0d0dull;
        const u64 nza = (((a & 0x7f7f7f7f7f7f7f7full) + 0x7f7f7f7f7f7f7f7full) | a) & 0x8080808080808080ull;   // high bit = byte != '\n'
        const u64 nzb = (((b & 0x7f7f7f7f7f7f7f7full) + 0x7f7f7f7f7f7f7f7full) | b) & 0x8080808080808080ull;
        breaks += __builtin_popcountll(~(nza & nzb) & 0x8080808080808080ull & keep);
        f += 8;
    }
    return c + (file_off - (d.seq_start + (rel & ~255ull))) - breaks;
}

// ---- reference stream ---------------------------------------------------------------------------
__global__ void k_fa_count(const uint8_t* __restrict__ f, u64 n, u64 seq_start, u64* __restrict__ blkcnt, u64 nblk)
{
    const u32 lane = threadIdx.x & 63;
    const u64 wave = (blockIdx.x * (u64)blockDim.x + threadIdx.x) >> 6, nw = ((u64)gridDim.x * blockDim.x) >> 6;
    for (u64 b = wave; b < nblk; b += nw) {
        const u64 base = seq_start + b * 256 + lane * 4;
        u32 c = 0;
        for (int i = 0; i < 4; i++) if (base + i < n) { uint8_t ch = f[base + i]; c += (ch != '\n' && ch != '\r'); }
        for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
        if (lane == 0) blkcnt[b] = c;
    }
}
__global__ void k_fa_compact(const uint8_t* __restrict__ f, u64 n, u64 seq_start, const u64* __restrict__ blkpre,
                             u64 nblk, uint8_t* __restrict__ refc)
{
    const u32 lane = threadIdx.x & 63;
    const u64 wave = (blockIdx.x * (u64)blockDim.x + threadIdx.x) >> 6, nw = ((u64)gridDim.x * blockDim.x) >> 6;
    for (u64 b = wave; b < nblk; b += nw) {
        const u64 base = seq_start + b * 256 + lane * 4;
        uint8_t ch[4];
        u32 c = 0;
        for (int i = 0; i < 4; i++) {
            ch[i] = base + i < n ? f[base + i] : (uint8_t)'\n';
            c += (ch[i] != '\n' && ch[i] != '\r');
        }
        u32 incl = c;
        for (int o = 1; o < 64; o <<= 1) { u32 a = __shfl_up(incl, o, 64); if (lane >= (u32)o) incl += a; }
        u64 o = blkpre[b] + incl - c;
        for (int i = 0; i < 4; i++) if (ch[i] != '\n' && ch[i] != '\r') refc[o++] = ch[i];
    }
}

// seq_size of parse_fasta_metadata (:69-84) on the device: the record ends at the first later line that starts with
// '>' (or at EOF); its size is the number of bytes that are not '\n' from the first sequence line on ('\r' counts,
// as in the reference's getline loop).
__global__ void k_fa_record_end(const uint8_t* __restrict__ f, u64 n, u64 from, u64* __restrict__ rec_end)
{
    u64 best = ~0ull;
    for (u64 i = from + blockIdx.x * (u64)blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x)
        if (f[i] == '>' && f[i - 1] == '\n') { best = i; break; }
    if (best != ~0ull) atomicMin((unsigned long long*)rec_end, (unsigned long long)best);
}
__global__ void k_fa_seq_bytes(const uint8_t* __restrict__ f, u64 from, const u64* __restrict__ rec_end, u64* __restrict__ count)
{
    const u64 hi = *rec_end;
    u64 c = 0;
    for (u64 i = from + blockIdx.x * (u64)blockDim.x + threadIdx.x; i < hi; i += (u64)gridDim.x * blockDim.x) c += f[i] != '\n';
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd((unsigned long long*)count, (unsigned long long)c);
}

// ---- grouping -------------------------------------------------------------------------------------
__global__ void k_rec_ends(const u64* __restrict__ start, const u64* __restrict__ reflen, u64 n, u64* __restrict__ ends)
{
    for (u64 j = blockIdx.x * (u64)blockDim.x + threadIdx.x; j < n; j += (u64)gridDim.x * blockDim.x)
        ends[j] = start[j] + reflen[j];
}
// a record opens a group iff its start is not below the largest end seen before it (:510)
__global__ void k_mark_groups(const u64* __restrict__ start, const u64* __restrict__ incl_end, u64 n, u64* __restrict__ flag)
{
    for (u64 j = blockIdx.x * (u64)blockDim.x + threadIdx.x; j < n; j += (u64)gridDim.x * blockDim.x)
        flag[j] = (j == 0) || !(start[j] < incl_end[j - 1]);
}
__global__ void k_group_firsts(const u64* __restrict__ flag, const u64* __restrict__ gidx, u64 n, u64* __restrict__ grp_r0,
                               const u64* __restrict__ ngrp)
{
    for (u64 j = blockIdx.x * (u64)blockDim.x + threadIdx.x; j < n; j += (u64)gridDim.x * blockDim.x)
        if (flag[j]) grp_r0[gidx[j]] = j;
    if (blockIdx.x == 0 && threadIdx.x == 0) grp_r0[*ngrp] = n;
}

// ---- per group: thread per group ---------------------------------------------------------------------
struct GrpArrays {
    u64* gs; u64* spanlen; u64* cs;          // span start (sequence position), clamped length, refc position
    u64* nraw; u64* rawchars;                // raw haplotypes (1 + sum of alts) and their total length
    u64* ndist;                              // distinct haplotypes
    u64* bitwords;                           // ndist * sample words
    u64* eds_len; u64* seds_len;             // group symbol + preceding common region
    u64* commonlen; u64* cur_after;          // common region before the group, reference position after it
    u64* err;                                // [0] first failing group, [1] offset, [2] span length
};

// apply_variant_to_span :356-390: span.substr(0, off) + alt + (off + |ref| < |span| ? span.substr(off + |ref|) : "").
// substr(0, n) never throws: an offset beyond the span (records at or below POS 0 with a long REF, groups
// past the end of the reference) just takes the whole span; off + reflen wraps like the reference's size_t.
__device__ __forceinline__ u64 hap_len(u64 off, u64 altlen, u64 reflen, u64 spanlen)
{
    u64 len = (off < spanlen ? off : spanlen) + altlen;
    if (off + reflen < spanlen) len += spanlen - (off + reflen);
    return len;
}

__global__ void k_grp_count(VcfDev d, GrpArrays a)
{
    for (u64 g = blockIdx.x * (u64)blockDim.x + threadIdx.x; g < d.ngrp; g += (u64)gridDim.x * blockDim.x) {
        const u64 r0 = d.grp_r0[g], r1 = d.grp_r0[g + 1];
        const u64 gs = d.start[r0], ge = d.incl_end[r1 - 1];
        u64 spanlen = 0, cs = d.refc_n;
        if (gs < d.seq_size) {                                 // read_fasta_region :102-109
            u64 length = ge - gs;
            if (gs + length > d.seq_size) length = d.seq_size - gs;
            cs = fa_cpos(d, d.seq_start + gs + gs / d.lw);
            spanlen = length < d.refc_n - cs ? length : d.refc_n - cs;
        }
        a.gs[g] = gs; a.spanlen[g] = spanlen; a.cs[g] = cs;
        u64 nraw = 1, chars = spanlen;
        for (u64 v = r0; v < r1; v++) {
            const u64 off = d.start[v] - gs;
            const u64 nalt = d.alt0[v + 1] - d.alt0[v];
            for (u64 x = d.alt0[v]; x < d.alt0[v + 1]; x++)
                chars += hap_len(off, d.altstr_off[x + 1] - d.altstr_off[x], d.reflen[v], spanlen);
            nraw += nalt;
        }
        a.nraw[g] = nraw; a.rawchars[g] = chars;
    }
}

struct HapArrays {
    const u64* raw0; const u64* rawc0;       // exclusive scans of nraw / rawchars
    u64* rawlen; u64* rawoff; u32* canon;    // per raw haplotype: length, offset in hapchars, canonical index
    uint8_t* hapchars;
    const u64* bit0;                         // exclusive scan of bitwords
    u64* carried;                            // bit matrix [canonical hap][sample word]
    u32 sw;                                  // sample words = ceil(max_samples / 64)
};

// materialise the raw haplotypes (span first, then every ALT applied alone: :416-435) and
// deduplicate them keeping the first occurrence
__global__ void k_grp_haps(VcfDev d, GrpArrays a, HapArrays h)
{
    for (u64 g = blockIdx.x * (u64)blockDim.x + threadIdx.x; g < d.ngrp; g += (u64)gridDim.x * blockDim.x) {
        const u64 r0 = d.grp_r0[g], r1 = d.grp_r0[g + 1];
        const u64 gs = a.gs[g], spanlen = a.spanlen[g], cs = a.cs[g];
        const uint8_t* span = d.refc + cs;
        u64 hi = h.raw0[g];
        u64 co = h.rawc0[g];
        // raw hap 0 = the reference span
        h.rawlen[hi] = spanlen; h.rawoff[hi] = co;
        for (u64 i = 0; i < spanlen; i++) h.hapchars[co + i] = span[i];
        co += spanlen; hi++;
        const u64 nraw = a.nraw[g];
        for (u64 v = r0; v < r1 && hi < h.raw0[g] + nraw; v++) {
            const u64 off = d.start[v] - gs, rl = d.reflen[v];
            for (u64 x = d.alt0[v]; x < d.alt0[v + 1]; x++) {
                const u64 al = d.altstr_off[x + 1] - d.altstr_off[x];
                const u64 len = hap_len(off, al, rl, spanlen);
                h.rawlen[hi] = len; h.rawoff[hi] = co;
                uint8_t* dst = h.hapchars + co;
                const u64 pre = off < spanlen ? off : spanlen;
                for (u64 i = 0; i < pre; i++) *dst++ = span[i];
                for (u64 i = 0; i < al; i++) *dst++ = d.altchars[d.altstr_off[x] + i];
                if (off + rl < spanlen) for (u64 i = off + rl; i < spanlen; i++) *dst++ = span[i];
                co += len; hi++;
            }
        }
        // dedup, first occurrence wins
        const u64 h0 = h.raw0[g];
        u32 nd = 0;
        for (u64 x = 0; x < nraw; x++) {
            u32 c = 0xffffffffu;
            for (u64 y = 0; y < x; y++) {
                if (h.rawlen[h0 + y] != h.rawlen[h0 + x] || h.canon[h0 + y] & 0x80000000u) continue;
                const uint8_t* p = h.hapchars + h.rawoff[h0 + x];
                const uint8_t* q = h.hapchars + h.rawoff[h0 + y];
                bool eq = true;
                for (u64 i = 0; i < h.rawlen[h0 + x]; i++) if (p[i] != q[i]) { eq = false; break; }
                if (eq) { c = h.canon[h0 + y]; break; }
            }
            // bit 31 marks "duplicate of an earlier one" so that only first occurrences are compared against
            h.canon[h0 + x] = c == 0xffffffffu ? nd++ : (c | 0x80000000u);
        }
        a.ndist[g] = nd;
        a.bitwords[g] = (u64)nd * h.sw;
    }
}

// which samples carry which haplotype (:438-473) and the text sizes (:570-651)
__global__ void k_grp_samples(VcfDev d, GrpArrays a, HapArrays h)
{
    for (u64 g = blockIdx.x * (u64)blockDim.x + threadIdx.x; g < d.ngrp; g += (u64)gridDim.x * blockDim.x) {
        const u64 r0 = d.grp_r0[g], r1 = d.grp_r0[g + 1];
        const u64 h0 = h.raw0[g];
        const u64 nd = a.ndist[g];
        u64* bits = h.carried + h.bit0[g];
        for (u64 i = 0; i < nd * h.sw; i++) bits[i] = 0;
        const u64 nsamp = d.pair0[r0 + 1] - d.pair0[r0];       // samples of the group's first record (:407)
        for (u64 s = 0; s < nsamp; s++) {
            bool any = false;
            u64 rawbase = 1;
            for (u64 v = r0; v < r1; v++) {
                const u64 nalt = d.alt0[v + 1] - d.alt0[v];
                const u64 ns_v = d.pair0[v + 1] - d.pair0[v];
                if (s < ns_v) {
                    const u64 pr = d.pair0[v] + s;
                    for (u64 t = d.pair_a0[pr]; t < d.pair_a0[pr + 1]; t++) {
                        const int al = d.alleles[t];
                        u32 idx = 0;                           // allele 0 or out of range -> the span
                        if (al >= 1 && (u64)al <= nalt) idx = h.canon[h0 + rawbase + (u64)(al - 1)] & 0x7fffffffu;
                        bits[(u64)idx * h.sw + (s >> 6)] |= 1ull << (s & 63);
                        any = true;
                    }
                }
                rawbase += nalt;
            }
            if (!any) bits[(s >> 6)] |= 1ull << (s & 63);      // no allele at all: reference (:466-468)
        }
        // sizes
        const u64 gs = a.gs[g], spanlen = a.spanlen[g];
        u64 eds = 2, seds = 0, nemit = 0;
        if (nsamp == 0) {                                      // no genotype columns: all haplotypes, one {0} (:603-615)
            for (u64 x = 0; x < a.nraw[g]; x++)
                if (!(h.canon[h0 + x] & 0x80000000u)) { eds += h.rawlen[h0 + x]; nemit++; }
            seds = 3;
        } else {
            for (u64 x = 0; x < a.nraw[g]; x++) {
                const u32 c = h.canon[h0 + x];
                if (c & 0x80000000u) continue;
                u64 cnt = 0, digits = 0;
                for (u64 s = 0; s < nsamp; s++)
                    if (bits[(u64)c * h.sw + (s >> 6)] >> (s & 63) & 1) { cnt++; digits += ndigits((u32)s + 1); }
                if (!cnt) continue;
                eds += h.rawlen[h0 + x]; nemit++;
                seds += 2 + digits + (cnt - 1);
            }
        }
        eds += nemit ? nemit - 1 : 0;
        a.eds_len[g] = eds; a.seds_len[g] = seds;
        a.cur_after[g] = gs + spanlen;                         // group.end_pos (:404)
    }
}

// the reference flushed before group g: [end of group g-1, gs) (:570-578); thread per group, after k_grp_samples
__global__ void k_grp_common(VcfDev d, GrpArrays a)
{
    for (u64 g = blockIdx.x * (u64)blockDim.x + threadIdx.x; g < d.ngrp; g += (u64)gridDim.x * blockDim.x) {
        const u64 cur = g ? a.cur_after[g - 1] : d.cur0;
        const u64 gs = a.gs[g];
        u64 clen = 0;
        if (gs > cur && cur < d.seq_size) {
            u64 length = gs - cur;
            if (cur + length > d.seq_size) length = d.seq_size - cur;
            const u64 c0 = fa_cpos(d, d.seq_start + cur + cur / d.lw);
            clen = length < d.refc_n - c0 ? length : d.refc_n - c0;
        }
        a.commonlen[g] = clen;
        if (clen) { a.eds_len[g] += clen + 2; a.seds_len[g] += 3; }
    }
}

struct EmitArrays { const u64* eds_off; const u64* seds_off; uint8_t* eds; uint8_t* seds; };

// common text in front of every group (the bandwidth part of the output).  A wave takes 64 consecutive groups: every
// lane fetches the numbers of one group (length, offsets, position in refc - one round of dependent loads for 64
// groups instead of one per group) and writes its brackets and "{0}"; then the wave copies the texts one group after the
// other, 16 bytes per lane (unaligned loads and stores), the ragged end byte by byte.
__device__ __forceinline__ u64 readlane64(u64 v, int l)
{
    const u32 lo = (u32)__builtin_amdgcn_readlane((int)(u32)v, l), hi = (u32)__builtin_amdgcn_readlane((int)(u32)(v >> 32), l);
    return (u64)lo | ((u64)hi << 32);
}
__global__ void __launch_bounds__(256) k_grp_emit_common(VcfDev d, GrpArrays a, EmitArrays e)
{
    const u32 lane = threadIdx.x & 63;
    const u64 wave = (blockIdx.x * (u64)blockDim.x + threadIdx.x) >> 6, nw = ((u64)gridDim.x * blockDim.x) >> 6;
    for (u64 gb = wave * 64; gb < d.ngrp; gb += nw * 64) {
        const u64 g = gb + lane;
        u64 clen = 0, eoff = 0, c0 = 0;
        if (g < d.ngrp) clen = a.commonlen[g];
        if (clen) {
            eoff = e.eds_off[g];
            uint8_t* eo = e.eds + eoff;
            uint8_t* so = e.seds + e.seds_off[g];
            const u64 cur = g ? a.cur_after[g - 1] : d.cur0;
            c0 = fa_cpos(d, d.seq_start + cur + cur / d.lw);
            eo[0] = '{'; eo[clen + 1] = '}'; so[0] = '{'; so[1] = '0'; so[2] = '}';
        }
        u64 todo = ballot64(clen != 0);
        while (todo) {
            const int l = __builtin_ctzll(todo);
            todo &= todo - 1;
            const u64 cl = readlane64(clen, l);
            uint8_t* dst = e.eds + readlane64(eoff, l) + 1;
            const uint8_t* src = d.refc + readlane64(c0, l);
            const u64 full = cl & ~15ull;
            for (u64 i = (u64)lane * 16; i < full; i += 1024) {
                const uint4 v = load16u(src + i);
                U128u w{v.x, v.y, v.z, v.w};
                __builtin_memcpy(dst + i, &w, 16);
            }
            if (lane < (u32)(cl - full)) dst[full + lane] = src[full + lane];
        }
    }
}

// the group symbols: short texts behind a chain of dependent loads, so one THREAD per group — a million independent
// chains in flight hide the latency that one lane per wave could not (3.7 ms -> see DESIGN section 5)
__global__ void __launch_bounds__(256) k_grp_emit(VcfDev d, GrpArrays a, HapArrays h, EmitArrays e)
{
    for (u64 g = blockIdx.x * (u64)blockDim.x + threadIdx.x; g < d.ngrp; g += (u64)gridDim.x * blockDim.x) {
        uint8_t* eo = e.eds + e.eds_off[g];
        uint8_t* so = e.seds + e.seds_off[g];
        const u64 clen = a.commonlen[g];
        if (clen) { eo += clen + 2; so += 3; }
        const u64 r0 = d.grp_r0[g];
        const u64 h0 = h.raw0[g];
        const u64 nsamp = d.pair0[r0 + 1] - d.pair0[r0];
        const u64* bits = h.carried + h.bit0[g];
        *eo++ = '{';
        bool first = true;
        for (u64 x = 0; x < a.nraw[g]; x++) {
            const u32 c = h.canon[h0 + x];
            if (c & 0x80000000u) continue;
            // carriers of this haplotype: one load per 64 samples (a per-sample load could not be kept in a register
            // across the byte stores below, which may alias it as far as the compiler knows)
            const u64* bw = bits + (u64)c * h.sw;
            const u32 nw = (u32)((nsamp + 63) >> 6);
            u64 cnt = 0;
            if (nsamp) {
                for (u32 w = 0; w < nw; w++) {
                    const u64 rem = nsamp - 64ull * w;
                    cnt += __builtin_popcountll(bw[w] & (rem >= 64 ? ~0ull : ((1ull << rem) - 1ull)));
                }
                if (!cnt) continue;
            }
            if (!first) *eo++ = ',';
            first = false;
            const uint8_t* src = h.hapchars + h.rawoff[h0 + x];
            for (u64 i = 0; i < h.rawlen[h0 + x]; i++) *eo++ = src[i];
            if (nsamp) {
                *so++ = '{';
                for (u32 w = 0; w < nw; w++) {
                    const u64 rem = nsamp - 64ull * w;
                    u64 v = bw[w] & (rem >= 64 ? ~0ull : ((1ull << rem) - 1ull));
                    while (v) {
                        const u32 id = w * 64u + (u32)__builtin_ctzll(v) + 1u, nd = ndigits(id);
                        v &= v - 1;
                        u32 xx = id;
                        for (int dd = (int)nd - 1; dd >= 0; dd--) { so[dd] = (uint8_t)('0' + xx % 10); xx /= 10; }
                        so += nd;
                        *so++ = ',';
                    }
                }
                so[-1] = '}';
            }
        }
        *eo++ = '}';
        if (!nsamp) { so[0] = '{'; so[1] = '0'; so[2] = '}'; }
    }
}

__global__ void k_cpos(VcfDev d, u64 file_off, u64* out) { *out = fa_cpos(d, file_off); }

// tail flush (:658-665)
__global__ void k_tail(VcfDev d, u64 cur, u64 clen, uint8_t* eo, uint8_t* so)
{
    const u64 c0 = fa_cpos(d, d.seq_start + cur + cur / d.lw);
    const u64 t = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (t == 0) { eo[0] = '{'; eo[clen + 1] = '}'; so[0] = '{'; so[1] = '0'; so[2] = '}'; }
    for (u64 i = t; i < clen; i += (u64)gridDim.x * blockDim.x) eo[1 + i] = d.refc[c0 + i];
}


// ---- device tokeniser (parse_vcf_line :232-326, parse_alt_field :142-176, parse_genotype :190-216) -------
// The VCF text goes to HBM as it is; record lines are found with a flag scan, a thread per record line walks its
// fields twice (count, then fill in sorted order) and writes the record SoA above — no host records, no uploads
// of eight arrays.  The kernels accept the *plain* spelling only: tab-separated lines without empty fields and
// with at least five of them, POS all digits (<= 19), no symbolic ALT other than <DEL>/<INS>, genotype alleles
// that are "." or all digits (<= 9), no '\r'.  Anything else — the whitespace-separated fallback of :262-279,
// malformed lines, unsupported structural variants (their warnings are in file order), signs or junk that
// std::stoull/stoi would swallow, POS 0 — raises `bad`, and the host tokeniser takes the whole file.
struct VtCtl { u64 n, bad, nrec, max_samples, t_alt, t_altc, t_pair, t_all, oob; };

struct VtSink {                      // fill pass: where record j's pieces go
    u64* altoff; u64 altc_base; uint8_t* altchars; u64* pa0; u64 all_base; int* alleles;
};
struct VtCounts { u64 pos; u64 reflen, nalt, altc, ngt, nall; };

// The bytes of the VCF text through 8-byte aligned loads.  A thread walks its line from left to right, so seven of
// eight byte reads come out of the register window instead of a one-byte load each (k_vt_count 0.80 -> 0.65 ms,
// k_vt_fill 0.96 -> 0.75 ms per 10^6 records, round 2).  Bounds: the text buffer starts 256-byte aligned (hipMalloc) and
// is allocated with at least 16 bytes of slack behind its n bytes (vt_raw_.ensure(n + 16)), so the aligned word that
// holds a byte i < n - the only bytes ever asked for - ends at most 7 bytes behind n, inside the allocation.
struct ByteWindow {
    const u64* words; u64 nbytes; u64 at = ~0ull; u64 v = 0; u64 oob = 0;
    __device__ __forceinline__ ByteWindow(const uint8_t* raw, u64 n) : words(reinterpret_cast<const u64*>(raw)), nbytes(n) {}
    __device__ __forceinline__ uint8_t operator[](u64 i)
    {
        if (i >= nbytes) { oob = i | (1ull << 63); return (uint8_t)'\n'; }   // never asked for by a correct walk: reported, not read
        const u64 wi = i >> 3;
        if (wi != at) { at = wi; v = words[wi]; }
        return (uint8_t)(v >> ((i & 7u) * 8u));
    }
};

template <bool FILL>
__device__ bool vt_parse(ByteWindow& raw, u64 lo, u64 hi, VtCounts& c, const VtSink& k)
{
    c = VtCounts{0, 0, 0, 0, 0, 0};
    u64 fs = lo, ref_lo = 0;
    int f = 0;
    for (u64 i = lo; i <= hi; i++) {
        if (i < hi && raw[i] != '\t') continue;
        const u64 fe = i;
        if (fe == fs) return false;                                      // empty token: the reference drops it (:262-268)
        if (f == 1) {
            if (fe - fs > 19) return false;
            u64 v = 0;
            for (u64 t = fs; t < fe; t++) { const uint8_t ch = raw[t]; if (ch < '0' || ch > '9') return false; v = v * 10 + (ch - '0'); }
            c.pos = v;
        } else if (f == 3) { ref_lo = fs; c.reflen = fe - fs; }
        else if (f == 4) {
            u64 t = fs;
            while (t < fe) {                                             // std::getline(ss, a, ',')
                u64 e = t;
                while (e < fe && raw[e] != ',') e++;
                const u64 len = e - t;
                u64 src = t;                                             // index of the allele's first byte in the text
                u64 alen = len;
                if (len && raw[t] == '<' && raw[e - 1] == '>') {
                    if (len == 5 && raw[t + 1] == 'D' && raw[t + 2] == 'E' && raw[t + 3] == 'L') alen = 0;
                    else if (len == 5 && raw[t + 1] == 'I' && raw[t + 2] == 'N' && raw[t + 3] == 'S') { src = ref_lo; alen = c.reflen; }
                    else return false;                                   // unsupported SV: warning + skip on the host
                }
                if (FILL) {
                    k.altoff[c.nalt] = k.altc_base + c.altc;
                    for (u64 x = 0; x < alen; x++) k.altchars[k.altc_base + c.altc + x] = raw[src + x];
                }
                c.nalt++; c.altc += alen;
                t = e + 1;
            }
        } else if (f >= 9) {
            u64 ge = fs;
            bool slash = false;
            while (ge < fe && raw[ge] != ':') { slash |= raw[ge] == '/'; ge++; }
            const uint8_t delim = slash ? '/' : '|';
            if (FILL) k.pa0[c.ngt] = k.all_base + c.nall;
            u64 t = fs;
            while (t < ge) {
                u64 e = t;
                while (e < ge && raw[e] != delim) e++;
                const u64 len = e - t;
                if (len && !(len == 1 && raw[t] == '.')) {
                    if (len > 9) return false;
                    int v = 0;
                    for (u64 x = t; x < e; x++) { const uint8_t ch = raw[x]; if (ch < '0' || ch > '9') return false; v = v * 10 + (ch - '0'); }
                    if (FILL) k.alleles[k.all_base + c.nall] = v;
                    c.nall++;
                }
                t = e + 1;
            }
            c.ngt++;
        }
        f++;
        fs = i + 1;
    }
    return f >= 5;
}

// Record-line starts without a flag and an index word per input BYTE (that scratch, 16 B per byte, sent VCFs of more than
// a few GB to the host tokeniser): a wave owns 1024 bytes of the text (16 per lane); pass 1 counts the line starts of
// every such block, a scan over the BLOCK counts (8 B per KB of text) numbers them, pass 2 finds them again and writes
// their positions.  A byte starts a record line iff it follows a newline (or is the first byte) and is neither a
// newline nor '#'.
constexpr u64 VT_BLOCK = 1024;
__device__ __forceinline__ u32 chunk_eq16b(const uint4& a, uint32_t cccc)      // bit i: byte i equals c
{
    return eq_byte4(a.x, cccc) | (eq_byte4(a.y, cccc) << 4) | (eq_byte4(a.z, cccc) << 8) | (eq_byte4(a.w, cccc) << 12);
}
__device__ __forceinline__ u32 vt_line_start_mask(const uint8_t* __restrict__ raw, u64 n, u64 i0, bool& saw_cr)
{
    if (i0 >= n) return 0;
    const uint4 v = *reinterpret_cast<const uint4*>(raw + i0);            // (the text buffer is 256-byte aligned, 16 bytes of slack)
    const u32 nl = chunk_eq16b(v, 0x0a0a0a0au), hash = chunk_eq16b(v, 0x23232323u);
    if (chunk_eq16b(v, 0x0d0d0d0du) & (n - i0 >= 16 ? 0xffffu : (1u << (n - i0)) - 1u)) saw_cr = true;
    const u32 prev_nl = ((nl << 1) | (i0 == 0 || raw[i0 - 1] == '\n' ? 1u : 0u)) & 0xffffu;
    u32 m = prev_nl & ~nl & ~hash;
    if (n - i0 < 16) m &= (1u << (n - i0)) - 1u;
    return m;
}
__global__ void __launch_bounds__(256) k_vt_line_count(const uint8_t* __restrict__ raw, u64 n, u64* __restrict__ cnt, VtCtl* ctl)
{
    const u64 nblk = (n + VT_BLOCK - 1) / VT_BLOCK;
    const u32 lane = threadIdx.x & 63;
    bool cr = false;
    for (u64 b = (blockIdx.x * (u64)blockDim.x + threadIdx.x) >> 6; b < nblk; b += ((u64)gridDim.x * blockDim.x) >> 6) {
        u32 c = (u32)__builtin_popcount(vt_line_start_mask(raw, n, b * VT_BLOCK + lane * 16u, cr));
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
        if (lane == 0) cnt[b] = c;
    }
    if (cr) ctl->bad = 1;
}
__global__ void __launch_bounds__(256) k_vt_line_fill(const uint8_t* __restrict__ raw, u64 n, const u64* __restrict__ base, u64* __restrict__ lstart)
{
    const u64 nblk = (n + VT_BLOCK - 1) / VT_BLOCK;
    const u32 lane = threadIdx.x & 63;
    bool cr = false;
    for (u64 b = (blockIdx.x * (u64)blockDim.x + threadIdx.x) >> 6; b < nblk; b += ((u64)gridDim.x * blockDim.x) >> 6) {
        const u64 i0 = b * VT_BLOCK + lane * 16u;
        u32 m = vt_line_start_mask(raw, n, i0, cr);
        const u32 c = (u32)__builtin_popcount(m);
        u32 incl = c;
        for (int o = 1; o < 64; o <<= 1) { const u32 x = __shfl_up(incl, o, 64); if (lane >= (u32)o) incl += x; }
        u64 at = base[b] + (incl - c);
        while (m) { lstart[at++] = i0 + (u32)__builtin_ctz(m); m &= m - 1; }
    }
}
__device__ __forceinline__ u64 vt_line_end(ByteWindow& raw, u64 lo, u64 n)
{
    u64 hi = lo;
    while (hi < n && raw[hi] != '\n') hi++;
    return hi;
}
struct VtRec { u64* pos; u64* reflen; u64* nalt; u64* altc; u64* ngt; u64* nall; u64* linelen; /* may be null */ };
__global__ void k_vt_count(const uint8_t* __restrict__ raw, u64 n, const u64* __restrict__ lstart, u64 nrec, VtRec r, VtCtl* ctl)
{
    bool bad = false;
    u64 mx = 0;
    for (u64 j = blockIdx.x * (u64)blockDim.x + threadIdx.x; j < nrec; j += (u64)gridDim.x * blockDim.x) {
        ByteWindow win(raw, n);
        const u64 lo = lstart[j], hi = vt_line_end(win, lo, n);
        VtCounts c;
        if (lo >= n) { ctl->oob = lo | (1ull << 62); bad = true; continue; }
        if (!vt_parse<false>(win, lo, hi, c, VtSink{})) bad = true;
        if (win.oob) ctl->oob = win.oob;
        if (r.linelen) r.linelen[j] = hi - lo;                               // index pass of a partitioned run
        else if (c.pos == 0 || c.pos - 1 + c.reflen < c.pos - 1) bad = true; // wrapped positions: host sweep (see run)
        r.pos[j] = c.pos; r.reflen[j] = c.reflen; r.nalt[j] = c.nalt; r.altc[j] = c.altc; r.ngt[j] = c.ngt; r.nall[j] = c.nall;
        mx = c.ngt > mx ? c.ngt : mx;
    }
    if (bad) ctl->bad = 1;
    if (mx) atomicMax((unsigned long long*)&ctl->max_samples, (unsigned long long)mx);
}
// strictly ascending positions: the reference's std::sort leaves such an array as it is (no two keys compare equal,
// so its result is THE sorted order) and the host sort, with its two PCIe trips, is skipped
__global__ void k_vt_ascending(const u64* __restrict__ pos, u64 nrec, VtCtl* ctl)
{
    bool un = false;
    for (u64 j = blockIdx.x * (u64)blockDim.x + threadIdx.x; j + 1 < nrec; j += (u64)gridDim.x * blockDim.x) un |= pos[j] >= pos[j + 1];
    if (un) ctl->t_alt = 1;                                      // scratch until the offset scans overwrite it
}
// ---- LSD radix sort of (u64 key, u32 value) pairs: eight stable passes of eight bits (unsorted VCFs with distinct
// positions only; the usual VCF ascends and skips it).  A wave owns a tile of RS_TILE consecutive elements: pass 1 counts
// its digits into a bin-major table (bin * ntiles + tile), one exclusive scan over the table gives every (bin, tile) its
// first output position, pass 2 walks the tile 64 elements at a time - lanes with the same digit find each other with
// one ballot per digit bit, their rank among them keeps the order stable - and moves the running positions in LDS.
constexpr u32 RS_TILE = 2048;
__global__ void __launch_bounds__(64) k_rs_hist(const u64* __restrict__ keys, u64 n, u32 shift, u64 ntiles, u64* __restrict__ table)
{
    __shared__ u32 hist[256];
    const u32 lane = threadIdx.x;
    for (u64 tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        for (u32 b = lane; b < 256; b += 64) hist[b] = 0;
        __syncthreads();
        const u64 base = tile * RS_TILE;
        for (u32 o = lane; o < RS_TILE; o += 64)
            if (base + o < n) atomicAdd(&hist[(u32)(keys[base + o] >> shift) & 0xffu], 1u);
        __syncthreads();
        for (u32 b = lane; b < 256; b += 64) table[(u64)b * ntiles + tile] = hist[b];
        __syncthreads();
    }
}
__global__ void __launch_bounds__(64) k_rs_scatter(const u64* __restrict__ keys, const u32* __restrict__ vals, u64 n, u32 shift,
                                                   u64 ntiles, const u64* __restrict__ table, u64* __restrict__ keys_out,
                                                   u32* __restrict__ vals_out)
{
    __shared__ u64 at[256];                                    // next output position of every digit of this tile
    const u32 lane = threadIdx.x;
    for (u64 tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        for (u32 b = lane; b < 256; b += 64) at[b] = table[(u64)b * ntiles + tile];
        __syncthreads();
        const u64 base = tile * RS_TILE;
        for (u32 o = 0; o < RS_TILE && base + o < n; o += 64) {
            const u64 i = base + o + lane;
            const bool valid = i < n;
            const u64 key = valid ? keys[i] : 0;
            const u32 val = valid ? (vals ? vals[i] : (u32)i) : 0u;         // vals == nullptr: the element's index (first pass)
            const u32 d = (u32)(key >> shift) & 0xffu;
            u64 same = ballot64(valid);                                      // lanes with this lane's digit
#pragma unroll
            for (int b = 0; b < 8; b++) {
                const u64 m = ballot64(valid && ((d >> b) & 1u));
                same &= ((d >> b) & 1u) ? m : ~m;
            }
            const u32 rank = mbcnt(same);
            if (valid) {
                const u64 pos = at[d] + rank;
                keys_out[pos] = key;
                vals_out[pos] = val;
            }
            __syncthreads();                                                 // (one wave: every lane has read at[] before it moves)
            if (valid && rank == 0) at[d] += (u64)__builtin_popcountll(same);
            __syncthreads();
        }
        __syncthreads();
    }
}
// counts in sorted order (inputs of the four offset scans)
__global__ void k_vt_gather(VtRec r, const u32* __restrict__ order, u64 nrec, u64* __restrict__ a, u64* __restrict__ b,
                            u64* __restrict__ c, u64* __restrict__ d)
{
    for (u64 j = blockIdx.x * (u64)blockDim.x + threadIdx.x; j < nrec; j += (u64)gridDim.x * blockDim.x) {
        const u64 i = order ? order[j] : j;
        a[j] = r.nalt[i]; b[j] = r.altc[i]; c[j] = r.ngt[i]; d[j] = r.nall[i];
    }
}
struct VtOut { u64* start; u64* reflen; u64* alt0; u64* altoff; uint8_t* altchars; u64* pair0; u64* pa0; int* alleles;
               const u64* altc0; const u64* gtc0; };
__global__ void k_vt_fill(const uint8_t* __restrict__ raw, u64 n, const u64* __restrict__ lstart, const u32* __restrict__ order,
                          u64 nrec, VtOut o, VtCtl* ctl)
{
    for (u64 j = blockIdx.x * (u64)blockDim.x + threadIdx.x; j < nrec; j += (u64)gridDim.x * blockDim.x) {
        const u64 lo = lstart[order ? order[j] : j];
        VtCounts c;
        VtSink k{o.altoff + o.alt0[j], o.altc0[j], o.altchars, o.pa0 + o.pair0[j], o.gtc0[j], o.alleles};
        ByteWindow win(raw, n);
        const u64 hi = vt_line_end(win, lo, n);
        if (lo < n) vt_parse<true>(win, lo, hi, c, k);
        else ctl->oob = lo | (1ull << 62);
        if (win.oob) ctl->oob = win.oob;
        o.start[j] = c.pos - 1;
        o.reflen[j] = c.reflen;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        o.alt0[nrec] = ctl->t_alt; o.altoff[ctl->t_alt] = ctl->t_altc;
        o.pair0[nrec] = ctl->t_pair; o.pa0[ctl->t_pair] = ctl->t_all;
    }
}

// ---- host ------------------------------------------------------------------------------------------
namespace {

bool next_line(const uint8_t* f, size_t n, size_t& pos, std::string& line, bool* hit_eof = nullptr)
{
    if (pos >= n) return false;
    const uint8_t* nl = static_cast<const uint8_t*>(memchr(f + pos, '\n', n - pos));
    size_t end = nl ? static_cast<size_t>(nl - f) : n;
    line.assign(reinterpret_cast<const char*>(f + pos), end - pos);
    if (hit_eof) *hit_eof = (nl == nullptr);
    pos = nl ? end + 1 : n;
    return true;
}

enum class Skip { NONE, HEADER, MALFORMED, UNSUPPORTED_SV };

// Records of one stretch of the file, flat (no allocation per record): record i has alts
// [alt0[i], alt0[i+1]) with characters altoff[..] .. altoff[..+1], and samples [gt0[i], gt0[i+1]) with
// alleles gtoff[..] .. gtoff[..+1].
struct VcfPart {
    std::vector<u64> pos, reflen;
    std::vector<u64> alt0{0}, altoff{0};
    std::vector<uint8_t> altchars;
    std::vector<u64> gt0{0}, gtoff{0};
    std::vector<int> alleles;
    std::vector<u64> loff, llen;                             // line of every accepted record (index pass only)
    VcfCounters st;
    std::vector<std::string> warns;
    size_t size() const { return pos.size(); }
    void drop_open_record()                                  // undo a record that turned out unsupported
    {
        altoff.resize(alt0.back() + 1); altchars.resize(altoff.back());
        gtoff.resize(gt0.back() + 1); alleles.resize(gtoff.back());
    }
};

struct Span { const char* p; size_t n; };

// parse_genotype :190-216 — tokens between delimiters as std::getline yields them (a trailing empty one
// is not produced), "." skipped, std::stoi decides what a number is (exceptions -> token ignored)
void parse_gt_flat(Span gt, VcfPart& out)
{
    const char delim = memchr(gt.p, '/', gt.n) ? '/' : '|';
    size_t i = 0;
    while (i < gt.n) {
        const char* e = static_cast<const char*>(memchr(gt.p + i, delim, gt.n - i));
        const size_t end = e ? static_cast<size_t>(e - gt.p) : gt.n;
        const size_t len = end - i;
        if (!(len == 1 && gt.p[i] == '.')) {
            if (len == 1 && gt.p[i] >= '0' && gt.p[i] <= '9') out.alleles.push_back(gt.p[i] - '0');
            else {
                try { out.alleles.push_back(std::stoi(std::string(gt.p + i, len))); } catch (...) {}
            }
        }
        i = end + 1;
    }
}

// parse_vcf_line :232-326 into the flat store; `fields` is scratch
Skip parse_line_flat(const char* line, size_t n, VcfPart& out, std::vector<Span>& fields, bool light = false)
{
    if (n == 0 || line[0] == '#') return Skip::HEADER;
    fields.clear();
    for (size_t i = 0; i < n;) {                             // std::getline(ss, tok, '\t'), empty tokens dropped
        const char* e = static_cast<const char*>(memchr(line + i, '\t', n - i));
        const size_t end = e ? static_cast<size_t>(e - line) : n;
        if (end > i) fields.push_back(Span{line + i, end - i});
        i = end + 1;
    }
    if (fields.size() < 5) {                                 // ws >> tok
        fields.clear();
        size_t i = 0;
        while (i < n) {
            while (i < n && std::isspace((unsigned char)line[i])) i++;
            size_t j = i;
            while (j < n && !std::isspace((unsigned char)line[j])) j++;
            if (j > i) fields.push_back(Span{line + i, j - i});
            i = j;
        }
    }
    if (fields.size() < 5) return Skip::MALFORMED;
    u64 pos;
    try { pos = std::stoull(std::string(fields[1].p, fields[1].n)); } catch (...) { return Skip::MALFORMED; }
    const Span ref = fields[3];
    // parse_alt_field :142-176 — std::getline(ss, a, ','): empty tokens kept, except one behind the last comma
    const Span alt = fields[4];
    for (size_t i = 0; i < alt.n;) {
        const char* e = static_cast<const char*>(memchr(alt.p + i, ',', alt.n - i));
        const size_t end = e ? static_cast<size_t>(e - alt.p) : alt.n;
        const char* a = alt.p + i;
        const size_t len = end - i;
        if (len && a[0] == '<' && a[len - 1] == '>') {
            const size_t tl = len >= 2 ? len - 2 : 0;        // "<" alone: substr(1, npos-ish) is empty in the reference too
            const std::string t = len >= 2 ? std::string(a + 1, tl) : std::string();
            if (t == "DEL") { out.altoff.push_back(out.altchars.size()); }
            else if (t == "INS") { out.altchars.insert(out.altchars.end(), ref.p, ref.p + ref.n); out.altoff.push_back(out.altchars.size()); }
            else {
                out.drop_open_record();
                out.warns.push_back("Warning: Skipping variant at " + std::string(fields[0].p, fields[0].n) + ":" +
                                    std::to_string(pos) + " - Unsupported structural variant type: " + t);
                return Skip::UNSUPPORTED_SV;
            }
        } else { out.altchars.insert(out.altchars.end(), a, a + len); out.altoff.push_back(out.altchars.size()); }
        i = end + 1;
    }
    if (fields.size() >= 10 && !light)
        for (size_t f = 9; f < fields.size(); f++) {
            Span gt = fields[f];
            const char* c = static_cast<const char*>(memchr(gt.p, ':', gt.n));
            if (c) gt.n = static_cast<size_t>(c - gt.p);
            parse_gt_flat(gt, out);
            out.gtoff.push_back(out.alleles.size());
        }
    out.pos.push_back(pos);
    out.reflen.push_back(ref.n);
    out.alt0.push_back(out.altoff.size() - 1);
    out.gt0.push_back(out.gtoff.size() - 1);
    return Skip::NONE;
}

template <class T> void upload(DevBuf& b, const std::vector<T>& v, hipStream_t st)
{
    b.ensure(sizeof(T) * (v.size() + 1) + 16);
    if (!v.empty()) EDSX_HIP(hipMemcpyAsync(b.ptr, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice, st));
}

// Lines are independent: large files are cut at line starts and tokenised by several host threads; records,
// counters and the stderr warnings are put together in file order, so the array handed to std::sort is the
// reference's.  light = positions, REF lengths and line spans only (the index pass of a partitioned run).
void tokenise(const uint8_t* vcf, size_t vcf_n, std::vector<VcfPart>& parts, std::vector<u64>& part_base,
              VcfCounters& stats, bool light)
{
    auto parse_range = [&](size_t lo, size_t hi, VcfPart& out) {
        std::vector<Span> fields;
        size_t pos = lo;
        while (pos < hi && pos < vcf_n) {
            const uint8_t* nl = static_cast<const uint8_t*>(memchr(vcf + pos, '\n', vcf_n - pos));
            const size_t end = nl ? static_cast<size_t>(nl - vcf) : vcf_n;
            const Skip skip = parse_line_flat(reinterpret_cast<const char*>(vcf + pos), end - pos, out, fields, light);
            if (skip == Skip::NONE) {
                out.st.total_variants++; out.st.processed_variants++;
                if (light) { out.loff.push_back(pos); out.llen.push_back(end - pos); }
            }
            else if (skip == Skip::MALFORMED) { out.st.total_variants++; out.st.skipped_malformed++; }
            else if (skip == Skip::UNSUPPORTED_SV) { out.st.total_variants++; out.st.skipped_unsupported_sv++; }
            pos = nl ? end + 1 : vcf_n;
        }
    };
    unsigned nt = 1;
    if (vcf_n >= ((size_t)4 << 20)) {
        const unsigned hc = std::thread::hardware_concurrency();
        nt = std::max(1u, std::min(16u, hc ? hc : 4u));
    }
    std::vector<size_t> cut(nt + 1, vcf_n);
    cut[0] = 0;
    for (unsigned t = 1; t < nt; t++) {                        // first line start at or after t*n/nt
        size_t g = (size_t)((unsigned __int128)vcf_n * t / nt);
        if (g < cut[t - 1]) g = cut[t - 1];
        const uint8_t* nl = g < vcf_n ? static_cast<const uint8_t*>(memchr(vcf + g, '\n', vcf_n - g)) : nullptr;
        cut[t] = nl ? static_cast<size_t>(nl - vcf) + 1 : vcf_n;
        if (g == 0) cut[t] = 0;                                // position 0 is a line start itself
    }
    parts.clear();
    parts.resize(nt);
    if (nt == 1) parse_range(0, vcf_n, parts[0]);
    else {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; t++) th.emplace_back([&, t] { parse_range(cut[t], cut[t + 1], parts[t]); });
        for (auto& x : th) x.join();
    }
    part_base.assign(nt + 1, 0);
    for (unsigned t = 0; t < nt; t++) {
        const VcfPart& pt = parts[t];
        part_base[t + 1] = part_base[t] + pt.size();
        stats.total_variants += pt.st.total_variants; stats.processed_variants += pt.st.processed_variants;
        stats.skipped_malformed += pt.st.skipped_malformed; stats.skipped_unsupported_sv += pt.st.skipped_unsupported_sv;
        for (const auto& w : pt.warns) fprintf(stderr, "%s\n", w.c_str());   // the reference warns on stderr (:301-302)
    }
    if (part_base[nt] >= 0xffffffffull) throw FormatError("VCF has too many records for this build");
}

// the reference's std::sort call (:715-718) on (pos, file index) pairs
void sort_like_reference(std::vector<std::pair<u64, u32>>& order)
{
    std::sort(order.begin(), order.end(),
              [](const std::pair<u64, u32>& a, const std::pair<u64, u32>& b) { return a.first < b.first; });
}

} // namespace

// ---- index pass and sort order of a partitioned run (VCF row) ------------------------------
void vcf_index(const uint8_t* vcf, size_t vcf_n, std::vector<u64>& pos, std::vector<u64>& reflen, std::vector<u64>& line_off,
               std::vector<u64>& line_len, VcfCounters& stats)
{
    stats = VcfCounters();
    std::vector<VcfPart> parts;
    std::vector<u64> part_base;
    tokenise(vcf, vcf_n, parts, part_base, stats, true);
    const size_t n = part_base.back();
    pos.clear(); reflen.clear(); line_off.clear(); line_len.clear();
    pos.reserve(n); reflen.reserve(n); line_off.reserve(n); line_len.reserve(n);
    for (const VcfPart& pt : parts) {
        pos.insert(pos.end(), pt.pos.begin(), pt.pos.end());
        reflen.insert(reflen.end(), pt.reflen.begin(), pt.reflen.end());
        line_off.insert(line_off.end(), pt.loff.begin(), pt.loff.end());
        line_len.insert(line_len.end(), pt.llen.begin(), pt.llen.end());
    }
}

void vcf_sort_order(const u64* pos, size_t n, u32* order_out)
{
    std::vector<std::pair<u64, u32>> order(n);
    for (size_t i = 0; i < n; i++) order[i] = {pos[i], (u32)i};
    sort_like_reference(order);
    for (size_t i = 0; i < n; i++) order_out[i] = order[i].second;
}

// Tokenise on the device (kernels above).  false: the file is not plain; the caller runs the host tokeniser.  On success
// the record SoA (sorted order) is in place in HBM and the counters are set.
bool VcfPipeline::tokenize_device(const uint8_t* vcf, size_t n, bool presorted, hipStream_t st, u64& nrec, u64& max_samples,
                                  VcfCounters& stats)
{
    { const char* e = getenv("EDSX_HOST_TOKENIZER"); if (e && atoi(e)) return false; }      // A/B switch for the parity tests
    nrec = 0; max_samples = 0;
    if (n == 0 || !device_scratch_fits(6 * n)) return false;   // raw text + the record arrays (a dozen u64 per record line)
    vt_raw_.ensure(n + 16);
    const u64 nblk = (n + VT_BLOCK - 1) / VT_BLOCK;
    vt_idx_.ensure(8 * (nblk + 2));                            // line starts per 1 KB block of the text
    scan_tmp_.ensure(8 * ((n + 2) / SCAN_TILE + 4));
    ctl_.ensure(8 * 32);
    VtCtl* ctl = reinterpret_cast<VtCtl*>(ctl_.as<u64>() + 16);
    VtCtl h{};
    h.n = nblk;                                                // element count of the block scan
    EDSX_HIP(hipMemcpyAsync(ctl, &h, sizeof(h), hipMemcpyHostToDevice, st));
    EDSX_HIP(hipMemcpyAsync(vt_raw_.ptr, vcf, n, hipMemcpyHostToDevice, st));
    const uint8_t* raw = vt_raw_.as<uint8_t>();
    hipLaunchKernelGGL(k_vt_line_count, dim3(2048), dim3(256), 0, st, raw, (u64)n, vt_idx_.as<u64>(), ctl);
    exclusive_scan_u64(vt_idx_.as<u64>(), vt_idx_.as<u64>(), &ctl->n, &ctl->nrec, scan_tmp_.as<u64>(), st);
    EDSX_HIP(hipMemcpyAsync(&h, ctl, sizeof(h), hipMemcpyDeviceToHost, st));
    EDSX_HIP(hipStreamSynchronize(st));
    if (h.bad || h.nrec >= 0xffffffffull) return false;
    const u64 nr = h.nrec;
    stats.total_variants = stats.processed_variants = nr;
    if (nr == 0) return true;
    vt_lstart_.ensure(8 * (nr + 1));
    hipLaunchKernelGGL(k_vt_line_fill, dim3(2048), dim3(256), 0, st, raw, (u64)n, vt_idx_.as<u64>(), vt_lstart_.as<u64>());
    for (DevBuf* b : {&vt_pos_, &vt_reflen_, &vt_nalt_, &vt_altc_, &vt_ngt_, &vt_nall_, &vt_s1_, &vt_s2_, &vt_s3_, &vt_s4_}) b->ensure(8 * (nr + 2));
    VtRec rec{vt_pos_.as<u64>(), vt_reflen_.as<u64>(), vt_nalt_.as<u64>(), vt_altc_.as<u64>(), vt_ngt_.as<u64>(), vt_nall_.as<u64>(), nullptr};
    hipLaunchKernelGGL(k_vt_count, dim3(2048), dim3(256), 0, st, raw, (u64)n, vt_lstart_.as<u64>(), nr, rec, ctl);
    hipLaunchKernelGGL(k_vt_ascending, dim3(1024), dim3(256), 0, st, vt_pos_.as<u64>(), nr, ctl);
    EDSX_HIP(hipMemcpyAsync(&h, ctl, sizeof(h), hipMemcpyDeviceToHost, st));
    EDSX_HIP(hipStreamSynchronize(st));
    if (h.oob) throw DeviceError("VCF tokeniser: byte index " + std::to_string(h.oob & ((1ull << 62) - 1)) + " outside the text of " +
                                 std::to_string(n) + " bytes (flags " + std::to_string(h.oob >> 62) + ")");
    if (h.bad) return false;
    // The reference's unstable std::sort (:715-718) on (pos, file index) pairs, as on the host path — unless the
    // positions already ascend strictly (the usual VCF) or the caller hands the records over in final order.
    const u32* d_order = nullptr;
    std::vector<u32> order;
    bool need_host_sort = !presorted && h.t_alt;
    if (need_host_sort) {
        // Not ascending.  If the positions are pairwise distinct there is exactly one sorted order, so a device radix
        // sort gives the reference's permutation; equal positions (checked on the sorted keys) need the host's
        // std::sort, whose unstable permutation of them is part of the output.
        vt_s2_.ensure(8 * (nr + 2)); vt_order_.ensure(4 * (nr + 1));
        {   // eight passes: positions -> tmp -> (vt_s2_, vt_order_) -> tmp ... the last pass ends in (vt_s2_, vt_order_)
            const u64 ntiles = (nr + RS_TILE - 1) / RS_TILE, nbins = 256 * ntiles;
            vt_sorttmp_.ensure(12 * (nr + 2) + 8 * (nbins + 2) + 64);
            u64* tkeys = vt_sorttmp_.as<u64>();
            u32* tvals = reinterpret_cast<u32*>(tkeys + nr + 2);
            u64* table = reinterpret_cast<u64*>(vt_sorttmp_.as<uint8_t>() + ((12 * (nr + 2) + 15) & ~(size_t)15));
            scan_tmp_.ensure(8 * (nbins / SCAN_TILE + 4));
            EDSX_HIP(hipMemcpyAsync(&ctl->n, &nbins, 8, hipMemcpyHostToDevice, st));
            const unsigned grid = (unsigned)std::min<u64>(ntiles, 1u << 16);
            for (u32 pass = 0; pass < 8; pass++) {
                const u64* kin = pass == 0 ? vt_pos_.as<u64>() : (pass & 1) ? tkeys : vt_s2_.as<u64>();
                const u32* vin = pass == 0 ? nullptr : (pass & 1) ? tvals : vt_order_.as<u32>();
                u64* kout = (pass & 1) ? vt_s2_.as<u64>() : tkeys;
                u32* vout = (pass & 1) ? vt_order_.as<u32>() : tvals;
                hipLaunchKernelGGL(k_rs_hist, dim3(grid), dim3(64), 0, st, kin, nr, 8 * pass, ntiles, table);
                exclusive_scan_u64(table, table, &ctl->n, &ctl->nrec, scan_tmp_.as<u64>(), st);
                hipLaunchKernelGGL(k_rs_scatter, dim3(grid), dim3(64), 0, st, kin, vin, nr, 8 * pass, ntiles, table, kout, vout);
            }
        }
        VtCtl z{};
        EDSX_HIP(hipMemcpyAsync(&ctl->t_alt, &z.t_alt, 8, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_vt_ascending, dim3(1024), dim3(256), 0, st, vt_s2_.as<u64>(), nr, ctl);   // sorted keys: strict?
        EDSX_HIP(hipMemcpyAsync(&h, ctl, sizeof(h), hipMemcpyDeviceToHost, st));
        EDSX_HIP(hipStreamSynchronize(st));
        if (!h.t_alt) { need_host_sort = false; d_order = vt_order_.as<u32>(); }
    }
    if (need_host_sort) {
        std::vector<u64> hpos(nr);
        EDSX_HIP(hipMemcpy(hpos.data(), vt_pos_.ptr, 8 * nr, hipMemcpyDeviceToHost));
        order.resize(nr);
        const auto ts0 = std::chrono::steady_clock::now();
        vcf_sort_order(hpos.data(), nr, order.data());
        { const char* e = getenv("EDSX_TRACE");
          if (e && atoi(e)) fprintf(stderr, "[edsx vcf]   std::sort of %llu positions %8.3f ms\n", (unsigned long long)nr,
                                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ts0).count()); }
        vt_order_.ensure(4 * (nr + 1));
        EDSX_HIP(hipMemcpyAsync(vt_order_.ptr, order.data(), 4 * nr, hipMemcpyHostToDevice, st));
        d_order = vt_order_.as<u32>();
    }
    h.n = nr;
    EDSX_HIP(hipMemcpyAsync(&ctl->n, &h.n, 8, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_vt_gather, dim3(2048), dim3(256), 0, st, rec, d_order, nr, vt_s1_.as<u64>(), vt_s2_.as<u64>(),
                       vt_s3_.as<u64>(), vt_s4_.as<u64>());
    alt0_.ensure(8 * (nr + 2)); pair0_.ensure(8 * (nr + 2)); start_.ensure(8 * (nr + 2)); reflen_.ensure(8 * (nr + 2));
    exclusive_scan_u64(vt_s1_.as<u64>(), alt0_.as<u64>(), &ctl->n, &ctl->t_alt, scan_tmp_.as<u64>(), st);
    exclusive_scan_u64(vt_s2_.as<u64>(), vt_s2_.as<u64>(), &ctl->n, &ctl->t_altc, scan_tmp_.as<u64>(), st);
    exclusive_scan_u64(vt_s3_.as<u64>(), pair0_.as<u64>(), &ctl->n, &ctl->t_pair, scan_tmp_.as<u64>(), st);
    exclusive_scan_u64(vt_s4_.as<u64>(), vt_s4_.as<u64>(), &ctl->n, &ctl->t_all, scan_tmp_.as<u64>(), st);
    EDSX_HIP(hipMemcpyAsync(&h, ctl, sizeof(h), hipMemcpyDeviceToHost, st));
    EDSX_HIP(hipStreamSynchronize(st));                          // also: `order` stays alive until its upload is done
    altoff_.ensure(8 * (h.t_alt + 2)); altchars_.ensure(h.t_altc + 16); pa0_.ensure(8 * (h.t_pair + 2)); alleles_.ensure(4 * (h.t_all + 4));
    VtOut o{start_.as<u64>(), reflen_.as<u64>(), alt0_.as<u64>(), altoff_.as<u64>(), altchars_.as<uint8_t>(), pair0_.as<u64>(),
            pa0_.as<u64>(), alleles_.as<int>(), vt_s2_.as<u64>(), vt_s4_.as<u64>()};
    hipLaunchKernelGGL(k_vt_fill, dim3(2048), dim3(256), 0, st, raw, (u64)n, vt_lstart_.as<u64>(), d_order, nr, o, ctl);
    EDSX_HIP(hipMemcpyAsync(&h, ctl, sizeof(h), hipMemcpyDeviceToHost, st));
    EDSX_HIP(hipStreamSynchronize(st));
    EDSX_HIP(hipGetLastError());
    if (h.oob) throw DeviceError("VCF tokeniser (fill): byte index " + std::to_string(h.oob & ((1ull << 62) - 1)) + " outside the text of " +
                                 std::to_string(n) + " bytes (flags " + std::to_string(h.oob >> 62) + ")");
    nrec = nr;
    max_samples = h.max_samples;
    return true;
}

// Index pass of a partitioned run on the device: the first half of tokenize_device (line starts, count pass) with the
// line lengths kept.  false: not a plain file, the host index pass takes it (wrapped positions are fine here: the
// planner decides what to do with them).
bool VcfPipeline::index_device(const uint8_t* vcf, size_t n, hipStream_t st, std::vector<u64>& pos, std::vector<u64>& reflen,
                               std::vector<u64>& line_off, std::vector<u64>& line_len, VcfCounters& stats)
{
    { const char* e = getenv("EDSX_HOST_TOKENIZER"); if (e && atoi(e)) return false; }
    if (n == 0 || !device_scratch_fits(6 * n)) return false;
    vt_raw_.ensure(n + 16);
    const u64 nblk = (n + VT_BLOCK - 1) / VT_BLOCK;
    vt_idx_.ensure(8 * (nblk + 2));                            // line starts per 1 KB block of the text
    scan_tmp_.ensure(8 * ((n + 2) / SCAN_TILE + 4));
    ctl_.ensure(8 * 32);
    VtCtl* ctl = reinterpret_cast<VtCtl*>(ctl_.as<u64>() + 16);
    VtCtl h{};
    h.n = nblk;                                                // element count of the block scan
    EDSX_HIP(hipMemcpyAsync(ctl, &h, sizeof(h), hipMemcpyHostToDevice, st));
    EDSX_HIP(hipMemcpyAsync(vt_raw_.ptr, vcf, n, hipMemcpyHostToDevice, st));
    const uint8_t* raw = vt_raw_.as<uint8_t>();
    hipLaunchKernelGGL(k_vt_line_count, dim3(2048), dim3(256), 0, st, raw, (u64)n, vt_idx_.as<u64>(), ctl);
    exclusive_scan_u64(vt_idx_.as<u64>(), vt_idx_.as<u64>(), &ctl->n, &ctl->nrec, scan_tmp_.as<u64>(), st);
    EDSX_HIP(hipMemcpyAsync(&h, ctl, sizeof(h), hipMemcpyDeviceToHost, st));
    EDSX_HIP(hipStreamSynchronize(st));
    if (h.bad || h.nrec >= 0xffffffffull) return false;
    const u64 nr = h.nrec;
    stats = VcfCounters();
    stats.total_variants = stats.processed_variants = nr;
    pos.assign(nr, 0); reflen.assign(nr, 0); line_off.assign(nr, 0); line_len.assign(nr, 0);
    if (nr == 0) return true;
    vt_lstart_.ensure(8 * (nr + 1));
    hipLaunchKernelGGL(k_vt_line_fill, dim3(2048), dim3(256), 0, st, raw, (u64)n, vt_idx_.as<u64>(), vt_lstart_.as<u64>());
    for (DevBuf* b : {&vt_pos_, &vt_reflen_, &vt_nalt_, &vt_altc_, &vt_ngt_, &vt_nall_, &vt_s1_}) b->ensure(8 * (nr + 2));
    VtRec rec{vt_pos_.as<u64>(), vt_reflen_.as<u64>(), vt_nalt_.as<u64>(), vt_altc_.as<u64>(), vt_ngt_.as<u64>(), vt_nall_.as<u64>(),
              vt_s1_.as<u64>()};
    hipLaunchKernelGGL(k_vt_count, dim3(2048), dim3(256), 0, st, raw, (u64)n, vt_lstart_.as<u64>(), nr, rec, ctl);
    EDSX_HIP(hipMemcpyAsync(pos.data(), vt_pos_.ptr, 8 * nr, hipMemcpyDeviceToHost, st));
    EDSX_HIP(hipMemcpyAsync(reflen.data(), vt_reflen_.ptr, 8 * nr, hipMemcpyDeviceToHost, st));
    EDSX_HIP(hipMemcpyAsync(line_off.data(), vt_lstart_.ptr, 8 * nr, hipMemcpyDeviceToHost, st));
    EDSX_HIP(hipMemcpyAsync(line_len.data(), vt_s1_.ptr, 8 * nr, hipMemcpyDeviceToHost, st));
    EDSX_HIP(hipMemcpyAsync(&h, ctl, sizeof(h), hipMemcpyDeviceToHost, st));
    EDSX_HIP(hipStreamSynchronize(st));
    EDSX_HIP(hipGetLastError());
    return !h.bad;
}

void VcfPipeline::run(const uint8_t* vcf, size_t vcf_n, const uint8_t* fasta, size_t fasta_n, HostBytes& eds,
                      HostBytes& seds, VcfCounters& stats, hipStream_t st, const VcfRange& range)
{
    // EDSX_TRACE=1: wall-clock of the host-visible stages on stderr (every mark follows a stream synchronisation)
    static const bool trace = [] { const char* e = getenv("EDSX_TRACE"); return e && atoi(e); }();
    auto t_last = std::chrono::steady_clock::now();
    auto mark = [&](const char* what) {
        if (!trace) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[edsx vcf] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    stats = VcfCounters();
    // ---- FASTA metadata (:51-86)
    u64 seq_start, lw, seq_size, rest_from = 0;
    {
        size_t pos = 0;
        std::string line;
        bool eof = false;
        if (!next_line(fasta, fasta_n, pos, line, &eof) || line.empty() || line[0] != '>')
            throw FormatError("Invalid FASTA format: expected header line starting with '>'");
        seq_start = eof ? fasta_n : pos;
        if (!next_line(fasta, fasta_n, pos, line)) throw FormatError("FASTA file is empty");
        lw = line.size();
        seq_size = line.size();                                // + the later lines, counted on the device (below)
        rest_from = pos;
    }
    mark("fasta metadata");
    // ---- VCF records (:690-712) and the unstable sort (:715-718)
    std::vector<VcfPart> parts;
    std::vector<u64> part_base;                              // first global record index of every part
    std::vector<std::pair<u64, u32>> order;                  // (pos, global index in file order), sorted by pos
    u64 dev_nrec = 0, dev_max_samples = 0;
    const bool on_device = tokenize_device(vcf, vcf_n, range.presorted, st, dev_nrec, dev_max_samples, stats);
    tokenised_on_device_ = on_device;
    mark(on_device ? "device tokenise" : "device tokenise attempt");
    if (!on_device) {
        stats = VcfCounters();
        tokenise(vcf, vcf_n, parts, part_base, stats, false);
        mark("host tokenise");
    }
    if (!on_device) {
        const unsigned nt = (unsigned)parts.size();
        // std::sort's permutation depends only on the outcomes of its comparisons, so sorting (pos, index)
        // pairs by pos ends in the order the reference's sort of whole records ends in
        order.resize(part_base[nt]);
        for (unsigned t = 0; t < nt; t++)
            for (size_t i = 0; i < parts[t].size(); i++) order[part_base[t] + i] = {parts[t].pos[i], (u32)(part_base[t] + i)};
        // a position range of a partitioned run hands its records over in their final order (the order of the
        // whole file's sort, edsx_vcf_sort_order): sorting the slice again could permute equal positions differently
        if (!range.presorted) sort_like_reference(order);
    }
    // record j of the sorted order -> (part, index in part)
    auto locate = [&](u64 j, const VcfPart*& pt) -> size_t {
        const u64 g = order[j].second;
        size_t t = std::upper_bound(part_base.begin(), part_base.end(), g) - part_base.begin() - 1;
        pt = &parts[t];
        return (size_t)(g - part_base[t]);
    };
    const u64 nrec = on_device ? dev_nrec : order.size();

    // a reference with an empty first line has no addressable positions; the reference divides by
    // line_width (UB), refuse instead
    if (lw == 0) throw FormatError("Invalid FASTA format: empty first sequence line");

    // ---- reference stream on the device
    d_fasta_.ensure(fasta_n + 16);
    EDSX_HIP(hipMemcpyAsync(d_fasta_.ptr, fasta, fasta_n, hipMemcpyHostToDevice, st));
    const u64 body = fasta_n > seq_start ? fasta_n - seq_start : 0;
    const u64 nblk = (body + 255) / 256;
    ctl_.ensure(8 * 32);
    u64* ctl = ctl_.as<u64>();
    blkpre_.ensure(8 * (nblk + 2));
    scan_tmp_.ensure(8 * ((std::max<u64>(nblk, nrec) + 2) / SCAN_TILE + 4));
    refc_.ensure(body + 16);
    u64 hctl[32] = {0};
    hctl[0] = nblk; hctl[10] = ~0ull;
    hctl[12] = fasta_n;                                      // record end (atomicMin), [13] bytes of the later lines
    EDSX_HIP(hipMemcpyAsync(ctl, hctl, sizeof(hctl), hipMemcpyHostToDevice, st));
    if (rest_from < fasta_n) {
        hipLaunchKernelGGL(k_fa_record_end, dim3(1024), dim3(256), 0, st, d_fasta_.as<uint8_t>(), (u64)fasta_n, rest_from, ctl + 12);
        hipLaunchKernelGGL(k_fa_seq_bytes, dim3(1024), dim3(256), 0, st, d_fasta_.as<uint8_t>(), rest_from, ctl + 12, ctl + 13);
    }
    if (nblk) {
        hipLaunchKernelGGL(k_fa_count, dim3(1024), dim3(256), 0, st, d_fasta_.as<uint8_t>(), (u64)fasta_n, seq_start,
                           blkpre_.as<u64>(), nblk);
        exclusive_scan_u64(blkpre_.as<u64>(), blkpre_.as<u64>(), ctl + 0, ctl + 1, scan_tmp_.as<u64>(), st);
        hipLaunchKernelGGL(k_fa_compact, dim3(1024), dim3(256), 0, st, d_fasta_.as<uint8_t>(), (u64)fasta_n, seq_start,
                           blkpre_.as<u64>(), nblk, refc_.as<uint8_t>());
    }
    EDSX_HIP(hipMemcpyAsync(hctl, ctl, sizeof(hctl), hipMemcpyDeviceToHost, st));
    EDSX_HIP(hipStreamSynchronize(st));
    const u64 refc_n = nblk ? hctl[1] : 0;
    seq_size += hctl[13];
    mark("fasta upload, metadata, compaction");

    VcfDev d{};
    d.fasta = d_fasta_.as<uint8_t>(); d.fasta_n = fasta_n; d.seq_start = seq_start; d.seq_size = seq_size; d.lw = lw;
    d.refc = refc_.as<uint8_t>(); d.refc_n = refc_n; d.blkpre = blkpre_.as<u64>();
    d.nrec = nrec; d.cur0 = range.cur0;

    u64 cur = range.cur0, ngrp = 0, E = 0, Q = 0;
    u64 max_samples = dev_max_samples;
    GrpArrays ga{};
    HapArrays ha{};
    if (nrec) {
        std::vector<u64> hstart, hreflen;                         // host path only (the device tokeniser refuses wrapped positions)
        if (!on_device) {
        // ---- records -> SoA
        // two passes: sizes -> offsets (serial prefix sums), then the fill by several host threads
        hstart.resize(nrec); hreflen.resize(nrec);
        std::vector<u64> halt0(nrec + 1), hpair0(nrec + 1);
        std::vector<u64> altc0(nrec + 1), gtc0(nrec + 1);        // first alt char / first allele of record j
        halt0[0] = 0; hpair0[0] = 0; altc0[0] = 0; gtc0[0] = 0;
        for (u64 j = 0; j < nrec; j++) {
            const VcfPart* pt;
            const size_t i = locate(j, pt);
            const u64 na = pt->alt0[i + 1] - pt->alt0[i], ng = pt->gt0[i + 1] - pt->gt0[i];
            halt0[j + 1] = halt0[j] + na;
            hpair0[j + 1] = hpair0[j] + ng;
            altc0[j + 1] = altc0[j] + (pt->altoff[pt->alt0[i + 1]] - pt->altoff[pt->alt0[i]]);
            gtc0[j + 1] = gtc0[j] + (pt->gtoff[pt->gt0[i + 1]] - pt->gtoff[pt->gt0[i]]);
            max_samples = std::max<u64>(max_samples, ng);
        }
        std::vector<u64> haltoff(halt0[nrec] + 1), hpa0(hpair0[nrec] + 1);
        std::vector<uint8_t> haltchars(altc0[nrec]);
        std::vector<int> halleles(gtc0[nrec]);
        haltoff[halt0[nrec]] = altc0[nrec];
        hpa0[hpair0[nrec]] = gtc0[nrec];
        {
            const unsigned hc = std::thread::hardware_concurrency();
            const unsigned ft = nrec >= 65536 ? std::max(1u, std::min(16u, hc ? hc : 4u)) : 1u;
            auto fill = [&](u64 lo, u64 hi) {
                for (u64 j = lo; j < hi; j++) {
                    const VcfPart* pt;
                    const size_t i = locate(j, pt);
                    hstart[j] = pt->pos[i] - 1;                    // wraps for POS 0 like the reference's size_t
                    hreflen[j] = pt->reflen[i];
                    const u64 a0 = pt->alt0[i], a1 = pt->alt0[i + 1], cbase = pt->altoff[a0];
                    for (u64 a = a0; a < a1; a++) haltoff[halt0[j] + (a - a0)] = altc0[j] + (pt->altoff[a] - cbase);
                    if (pt->altoff[a1] > cbase) memcpy(haltchars.data() + altc0[j], pt->altchars.data() + cbase, pt->altoff[a1] - cbase);
                    const u64 g0 = pt->gt0[i], g1 = pt->gt0[i + 1], ebase = pt->gtoff[g0];
                    for (u64 g = g0; g < g1; g++) hpa0[hpair0[j] + (g - g0)] = gtc0[j] + (pt->gtoff[g] - ebase);
                    if (pt->gtoff[g1] > ebase)
                        memcpy(halleles.data() + gtc0[j], pt->alleles.data() + ebase, sizeof(int) * (pt->gtoff[g1] - ebase));
                }
            };
            if (ft == 1) fill(0, nrec);
            else {
                std::vector<std::thread> th;
                for (unsigned t = 0; t < ft; t++) th.emplace_back(fill, nrec * t / ft, nrec * (t + 1) / ft);
                for (auto& x : th) x.join();
            }
        }
        upload(start_, hstart, st); upload(reflen_, hreflen, st); upload(alt0_, halt0, st); upload(altoff_, haltoff, st);
        upload(altchars_, haltchars, st); upload(pair0_, hpair0, st); upload(pa0_, hpa0, st); upload(alleles_, halleles, st);
        }
        d.start = start_.as<u64>(); d.reflen = reflen_.as<u64>(); d.alt0 = alt0_.as<u64>(); d.altstr_off = altoff_.as<u64>();
        d.altchars = altchars_.as<uint8_t>(); d.pair0 = pair0_.as<u64>(); d.pair_a0 = pa0_.as<u64>(); d.alleles = alleles_.as<int>();

        // ---- groups
        ends_.ensure(8 * (nrec + 2)); flag_.ensure(8 * (nrec + 2)); gidx_.ensure(8 * (nrec + 2)); grp_r0_.ensure(8 * (nrec + 2));
        hctl[0] = nrec;
        EDSX_HIP(hipMemcpyAsync(ctl, hctl, 8, hipMemcpyHostToDevice, st));
        // The parallel rule "a record opens a group iff its start is not below the largest end seen so far" equals
        // the reference's sweep (:482-534, group_end restarts with every group) as long as starts ascend with the
        // sort key and no end wraps.  POS 0 (start = 2^64 - 1) or POS near 2^64 ("-5" parses!) break that; such
        // files are swept on the host exactly as the reference does it (running group end per record).
        bool wraps = false;
        for (u64 j = 0; j < hstart.size() && !wraps; j++) wraps = hstart[j] == ~0ull || hstart[j] + hreflen[j] < hstart[j];
        if (!wraps) {
            hipLaunchKernelGGL(k_rec_ends, dim3(1024), dim3(256), 0, st, d.start, d.reflen, nrec, ends_.as<u64>());
            inclusive_max_scan_u64(ends_.as<u64>(), ends_.as<u64>(), ctl + 0, ctl + 2, scan_tmp_.as<u64>(), st);
            hipLaunchKernelGGL(k_mark_groups, dim3(1024), dim3(256), 0, st, d.start, ends_.as<u64>(), nrec, flag_.as<u64>());
        } else {
            std::vector<u64> hflag(nrec), hend(nrec);
            u64 ge = 0;
            for (u64 j = 0; j < nrec; j++) {
                const u64 e = hstart[j] + hreflen[j];
                if (j == 0 || !(hstart[j] < ge)) { hflag[j] = 1; ge = e; }
                else { hflag[j] = 0; ge = std::max(ge, e); }
                hend[j] = ge;
            }
            EDSX_HIP(hipMemcpyAsync(flag_.ptr, hflag.data(), 8 * nrec, hipMemcpyHostToDevice, st));
            EDSX_HIP(hipMemcpyAsync(ends_.ptr, hend.data(), 8 * nrec, hipMemcpyHostToDevice, st));
            EDSX_HIP(hipStreamSynchronize(st));                    // the vectors go out of scope
        }
        exclusive_scan_u64(flag_.as<u64>(), gidx_.as<u64>(), ctl + 0, ctl + 3, scan_tmp_.as<u64>(), st);
        hipLaunchKernelGGL(k_group_firsts, dim3(1024), dim3(256), 0, st, flag_.as<u64>(), gidx_.as<u64>(), nrec,
                           grp_r0_.as<u64>(), ctl + 3);
        EDSX_HIP(hipMemcpyAsync(hctl, ctl, sizeof(hctl), hipMemcpyDeviceToHost, st));
        EDSX_HIP(hipStreamSynchronize(st));
        ngrp = hctl[3];
        d.ngrp = ngrp; d.grp_r0 = grp_r0_.as<u64>(); d.incl_end = ends_.as<u64>();

        // ---- per group
        for (DevBuf* b : {&g_gs_, &g_spanlen_, &g_cs_, &g_nraw_, &g_rawchars_, &g_ndist_, &g_bitwords_, &g_eds_, &g_seds_,
                          &g_common_, &g_cur_}) b->ensure(8 * (ngrp + 2));
        ga = GrpArrays{g_gs_.as<u64>(), g_spanlen_.as<u64>(), g_cs_.as<u64>(), g_nraw_.as<u64>(), g_rawchars_.as<u64>(),
                     g_ndist_.as<u64>(), g_bitwords_.as<u64>(), g_eds_.as<u64>(), g_seds_.as<u64>(), g_common_.as<u64>(),
                       g_cur_.as<u64>(), ctl + 10};
        scan_tmp_.ensure(8 * ((ngrp + 2) / SCAN_TILE + 4));
        raw0_.ensure(8 * (ngrp + 2)); rawc0_.ensure(8 * (ngrp + 2)); bit0_.ensure(8 * (ngrp + 2));
        hctl[0] = ngrp;
        EDSX_HIP(hipMemcpyAsync(ctl, hctl, 8, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_grp_count, dim3(1024), dim3(256), 0, st, d, ga);
        exclusive_scan_u64(ga.nraw, raw0_.as<u64>(), ctl + 0, ctl + 4, scan_tmp_.as<u64>(), st);
        exclusive_scan_u64(ga.rawchars, rawc0_.as<u64>(), ctl + 0, ctl + 5, scan_tmp_.as<u64>(), st);
        EDSX_HIP(hipMemcpyAsync(hctl, ctl, sizeof(hctl), hipMemcpyDeviceToHost, st));
        EDSX_HIP(hipStreamSynchronize(st));
        const u64 nraw_total = hctl[4], rawchars_total = hctl[5];
        const u32 sw = (u32)std::max<u64>(1, (max_samples + 63) / 64);
        rawlen_.ensure(8 * (nraw_total + 2)); rawoff_.ensure(8 * (nraw_total + 2)); canon_.ensure(4 * (nraw_total + 2));
        hapchars_.ensure(rawchars_total + 16);
        ha = HapArrays{raw0_.as<u64>(), rawc0_.as<u64>(), rawlen_.as<u64>(), rawoff_.as<u64>(), canon_.as<u32>(),
                       hapchars_.as<uint8_t>(), bit0_.as<u64>(), nullptr, sw};
        hipLaunchKernelGGL(k_grp_haps, dim3(1024), dim3(256), 0, st, d, ga, ha);
        exclusive_scan_u64(ga.bitwords, bit0_.as<u64>(), ctl + 0, ctl + 6, scan_tmp_.as<u64>(), st);
        EDSX_HIP(hipMemcpyAsync(hctl, ctl, sizeof(hctl), hipMemcpyDeviceToHost, st));
        EDSX_HIP(hipStreamSynchronize(st));
        carried_.ensure(8 * (hctl[6] + 2));
        ha.carried = carried_.as<u64>();
        hipLaunchKernelGGL(k_grp_samples, dim3(1024), dim3(256), 0, st, d, ga, ha);
        hipLaunchKernelGGL(k_grp_common, dim3(1024), dim3(256), 0, st, d, ga);
        exclusive_scan_u64(ga.eds_len, ga.eds_len, ctl + 0, ctl + 7, scan_tmp_.as<u64>(), st);
        exclusive_scan_u64(ga.seds_len, ga.seds_len, ctl + 0, ctl + 8, scan_tmp_.as<u64>(), st);
        EDSX_HIP(hipMemcpyAsync(hctl, ctl, sizeof(hctl), hipMemcpyDeviceToHost, st));
        u64 last_cur = 0;
        EDSX_HIP(hipMemcpyAsync(&last_cur, g_cur_.as<u64>() + (ngrp - 1), 8, hipMemcpyDeviceToHost, st));
        EDSX_HIP(hipStreamSynchronize(st));
        E = hctl[7]; Q = hctl[8]; cur = last_cur;
        mark("groups, haplotypes, sizes");
    }
    // ---- tail (:658-665)
    // A position range that is not the last one ends with what the reference flushes in front of the next
    // range's first group instead (:570-578, `start > cur`, text clamped by read_fasta_region).
    const bool last_range = range.next_start == ~0ull;
    u64 tail = 0;
    if (cur < seq_size && (last_range || range.next_start > cur)) {
        // clamp exactly like read_fasta_region: by seq_size and by what the file still holds
        u64 length = last_range ? seq_size - cur : std::min<u64>(range.next_start - cur, seq_size - cur);
        // position of `cur` in the newline-free stream: ask the device map (one thread)
        hipLaunchKernelGGL(k_cpos, dim3(1), dim3(1), 0, st, d, seq_start + cur + cur / lw, ctl + 11);
        u64 c0 = 0;
        EDSX_HIP(hipMemcpyAsync(&c0, ctl + 11, 8, hipMemcpyDeviceToHost, st));
        EDSX_HIP(hipStreamSynchronize(st));
        tail = std::min<u64>(length, refc_n - c0);
    }
    const u64 Etot = E + (tail ? tail + 2 : 0), Qtot = Q + (tail ? 3 : 0);
    d_eds_.ensure(Etot + 16); d_seds_.ensure(Qtot + 16);
    if (ngrp) {
        EmitArrays ea{g_eds_.as<u64>(), g_seds_.as<u64>(), d_eds_.as<uint8_t>(), d_seds_.as<uint8_t>()};
        hipLaunchKernelGGL(k_grp_emit_common, dim3(2048), dim3(256), 0, st, d, ga, ea);
        hipLaunchKernelGGL(k_grp_emit, dim3(2048), dim3(256), 0, st, d, ga, ha, ea);
    }
    if (tail)
        hipLaunchKernelGGL(k_tail, dim3(1024), dim3(256), 0, st, d, cur, tail, d_eds_.as<uint8_t>() + E, d_seds_.as<uint8_t>() + Q);
    mark("tail");
    // malloc'ed, not value-initialised: the pages are first touched by the copy itself (a zero-filling
    // std::string::resize costs as much as the whole device part for a 134 MB output)
    eds.take(Etot);
    seds.take(Qtot);
    PinnedDownload::copy(eds.data, d_eds_.ptr, Etot, st);
    PinnedDownload::copy(seds.data, d_seds_.ptr, Qtot, st);
    EDSX_HIP(hipStreamSynchronize(st));
    EDSX_HIP(hipGetLastError());
    stats.variant_groups = ngrp;                             // :724-726
    mark("emit + download");
}

} // namespace edsx
