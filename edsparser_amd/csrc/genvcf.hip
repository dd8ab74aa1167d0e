// genvcf.hip — synthetic VCF + FASTA of BASELINE configs[3]'s shape, generated in HBM: one FASTA record of
// `ref_len` uniform ACGT bases in 60-column lines; `n_records` record lines with strictly ascending, distinct POS (one per
// stride of ref_len / n_records positions), 70 % SNP / 15 % insertion of 1..10 bases / 15 % deletion of 1..10 bases (REF
// spans the deleted bases, so a few per cent of the records overlap the next POS and exercise the grouping of
// vcf_transforms.cpp:482-534), `n_samples` diploid phased samples with each allele ALT at p = 0.3, tab-separated, GT only.
// Counter-based (splitmix hashes of (seed, index)): the text depends on the parameters only.  Three passes over the
// records: line lengths, their exclusive scan, the text.
#include "genvcf.hpp"

namespace edsx {

namespace {

__device__ __forceinline__ uint8_t ref_base(u64 seed, u64 p) { return "ACGT"[hash3(seed, p, 0x51) & 3]; }   // 0-based position

struct VRec { u64 pos; u32 kind, len; };           // 1-based POS; 0 SNP, 1 insertion of len bases, 2 deletion of len bases
__device__ __forceinline__ VRec vrec(u64 seed, u64 i, u64 stride, u64 ref_len)
{
    VRec r;
    const u64 h = hash3(seed, i, 0x77);
    r.pos = 1 + i * stride + (h >> 20) % stride;
    if (r.pos + 12 > ref_len) r.pos = ref_len > 12 ? ref_len - 12 : 1;       // (only the last stride can reach the end)
    const u32 u = (u32)h & 0xfffffu;
    r.kind = u < 734003u ? 0u : u < 891290u ? 1u : 2u;                       // 70 % / 15 % / 15 % of 2^20
    r.len = 1 + (u32)((h >> 44) % 10);
    return r;
}
__device__ __forceinline__ u32 ndig64(u64 v) { u32 d = 1; while (v >= 10) { v /= 10; d++; } return d; }
// "chr1\t" POS "\t.\t" REF "\t" ALT "\t.\tPASS\t.\tGT" + n_samples x "\ta|b" + "\n"
__device__ __forceinline__ u64 vrec_len(const VRec& r, u32 ns)
{
    const u64 refl = r.kind == 2 ? 1 + r.len : 1, altl = r.kind == 1 ? 1 + r.len : 1;
    return 5 + ndig64(r.pos) + 3 + refl + 1 + altl + 12 + 4ull * ns + 1;
}

__global__ void k_gv_len(u64 seed, u64 n, u64 stride, u64 ref_len, u32 ns, u64* __restrict__ len)
{
    for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x)
        len[i] = vrec_len(vrec(seed, i, stride, ref_len), ns);
}
__device__ __forceinline__ uint8_t* put(uint8_t* o, const char* s) { while (*s) *o++ = (uint8_t)*s++; return o; }
__global__ void k_gv_fill(u64 seed, u64 n, u64 stride, u64 ref_len, u32 ns, const u64* __restrict__ off, u64 hdr_len,
                          uint8_t* __restrict__ out)
{
    for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        const VRec r = vrec(seed, i, stride, ref_len);
        uint8_t* o = out + hdr_len + off[i];
        o = put(o, "chr1\t");
        { const u32 nd = ndig64(r.pos); u64 v = r.pos; for (int k = (int)nd - 1; k >= 0; k--) { o[k] = (uint8_t)('0' + v % 10); v /= 10; } o += nd; }
        o = put(o, "\t.\t");
        const uint8_t base = ref_base(seed, r.pos - 1);
        *o++ = base;
        if (r.kind == 2) for (u32 k = 0; k < r.len; k++) *o++ = ref_base(seed, r.pos + k);
        *o++ = '\t';
        if (r.kind == 0) *o++ = "ACGT"[((hash3(seed, r.pos - 1, 0x51) & 3) + 1 + hash3(seed, i, 0x78) % 3) & 3];
        else {
            *o++ = base;
            if (r.kind == 1) for (u32 k = 0; k < r.len; k++) *o++ = "ACGT"[hash3(seed, i, 0x100 + k) & 3];
        }
        o = put(o, "\t.\tPASS\t.\tGT");
        for (u32 smp = 0; smp < ns; smp++) {
            const u64 g = hash3(seed, i, 0x1000 + smp);
            *o++ = '\t'; *o++ = (g & 0xffffu) < 19661u ? '1' : '0'; *o++ = '|'; *o++ = ((g >> 16) & 0xffffu) < 19661u ? '1' : '0';
        }
        *o = '\n';
    }
}
// FASTA: ">chr1 synthetic\n" + 60-column lines
__global__ void k_gv_fasta(u64 seed, u64 ref_len, u64 hdr_len, u64 total, uint8_t* __restrict__ out)
{
    for (u64 b = blockIdx.x * (u64)blockDim.x + threadIdx.x; b < total - hdr_len; b += (u64)gridDim.x * blockDim.x) {
        const u64 line = b / 61, col = b % 61;
        const u64 p = line * 60 + col;
        out[hdr_len + b] = (col == 60 || p >= ref_len) ? (uint8_t)'\n' : ref_base(seed, p);
    }
}

} // namespace

void GenVcfPipeline::run(u64 ref_len, u64 n_records, u32 n_samples, u64 seed, HostBytes& vcf, HostBytes& fasta, hipStream_t st)
{
    if (ref_len < 64 || n_records == 0 || n_records > ref_len / 16) throw ParamError("genvcf: need ref_len >= 64 and 1 <= n_records <= ref_len / 16");
    if (n_samples > 4096) throw ParamError("genvcf: at most 4096 samples");
    const u64 stride = ref_len / n_records;
    std::string hdr = "##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT";
    for (u32 i = 0; i < n_samples; i++) hdr += "\ts" + std::to_string(i);
    hdr += "\n";
    len_.ensure(8 * (n_records + 2)); scan_tmp_.ensure(8 * ((n_records + 2) / SCAN_TILE + 4)); ctl_.ensure(64);
    u64* ctl = ctl_.as<u64>();
    u64 h[2] = {n_records, 0};
    EDSX_HIP(hipMemcpyAsync(ctl, h, sizeof(h), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_gv_len, dim3(2048), dim3(256), 0, st, seed, n_records, stride, ref_len, n_samples, len_.as<u64>());
    exclusive_scan_u64(len_.as<u64>(), len_.as<u64>(), &ctl[0], &ctl[1], scan_tmp_.as<u64>(), st);
    EDSX_HIP(hipMemcpyAsync(h, ctl, sizeof(h), hipMemcpyDeviceToHost, st));
    EDSX_HIP(hipStreamSynchronize(st));
    const u64 vcf_bytes = hdr.size() + h[1];
    out_.ensure(vcf_bytes + 16);
    EDSX_HIP(hipMemcpyAsync(out_.ptr, hdr.data(), hdr.size(), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_gv_fill, dim3(2048), dim3(256), 0, st, seed, n_records, stride, ref_len, n_samples, len_.as<u64>(),
                       (u64)hdr.size(), out_.as<uint8_t>());
    EDSX_HIP(hipGetLastError());
    vcf.take(vcf_bytes);
    PinnedDownload::copy(vcf.data, out_.ptr, vcf_bytes, st);
    const std::string fh = ">chr1 synthetic\n";
    const u64 nlines = (ref_len + 59) / 60, fa_bytes = fh.size() + ref_len + nlines;
    out_.ensure(fa_bytes + 16);
    EDSX_HIP(hipMemcpyAsync(out_.ptr, fh.data(), fh.size(), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_gv_fasta, dim3(4096), dim3(256), 0, st, seed, ref_len, (u64)fh.size(), fa_bytes, out_.as<uint8_t>());
    EDSX_HIP(hipGetLastError());
    fasta.take(fa_bytes);
    PinnedDownload::copy(fasta.data, out_.ptr, fa_bytes, st);
}

} // namespace edsx
