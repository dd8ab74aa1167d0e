// msa_scan.hip - translation unit of the row index (K0), the column scan with its fused grouping (K1) and the
// runs -> segments kernels (K2) of the MSA -> EDS engine, and their launchers (msa_scan_launch.hpp).  Built with
// `-mllvm -enable-misched=false` (edsparser_amd/build.py): see msa_scan_launch.hpp.
#include "msa_scan_kernels.hpp"

namespace edsx {

void launch_find_hdr_end(const uint8_t* f, u64 n, MsaHdr* h, hipStream_t st)
{
    hipLaunchKernelGGL(k_find_hdr_end, dim3(1), dim3(64), 0, st, f, n, h);
}
void launch_find_row0(const uint8_t* f, u64 n, MsaHdr* h, hipStream_t st)
{
    hipLaunchKernelGGL(k_find_row0, dim3(512), dim3(1024), 0, st, f, n, h);
}
void launch_index_spec(const uint8_t* f, u64 n, const MsaHdr* h, u64* hpos, u64* cand, u64 row_cap, hipStream_t st)
{
    hipLaunchKernelGGL(k_index_spec, dim3(512), dim3(256), 0, st, f, n, h, hpos, cand, row_cap);
}
void launch_index_check(const uint8_t* f, u64 n, MsaHdr* h, const u64* hpos, const u64* cand, u64* row_start, u64 row_cap, hipStream_t st)
{
    hipLaunchKernelGGL(k_index_check, dim3(1), dim3(1024), 0, st, f, n, h, hpos, cand, row_start, row_cap);
}
void launch_index_rows(const uint8_t* f, u64 n, MsaHdr* h, u64* row_start, u64 row_cap, hipStream_t st)
{
    hipLaunchKernelGGL(k_index_rows, dim3(1), dim3(64), 0, st, f, n, h, row_start, row_cap);
}
void launch_pad_rows(u64* row_start, u64 S, u64 n, hipStream_t st)
{
    hipLaunchKernelGGL(k_pad_rows, dim3(4), dim3(256), 0, st, row_start, S, n);
}

template <int T, int RPT, bool HOLD, bool LANEROWS, int MINW, bool ROWS64 = false, bool BIG = false>
static void launch_k1(const K1Params& p, size_t lds, hipStream_t st)
{
    auto kern = k_scan_extract<T, RPT, HOLD, LANEROWS, MINW, ROWS64, BIG>;
    EDSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)p.ntiles), dim3(T), lds, st, p);
}
void launch_scan_extract(const K1Params& p, int threads, bool hold, bool lane_rows, bool rows64, bool big, size_t lds, hipStream_t st)
{
    if (lane_rows && rows64) launch_k1<512, 16, true, true, 4, true>(p, lds, st);    // one row per lane in the fused grouping
    else if (lane_rows && threads == 1024) launch_k1<1024, 16, true, true, 4>(p, lds, st);
    else if (lane_rows) launch_k1<512, 16, true, true, 4>(p, lds, st);
    else if (hold) launch_k1<512, 16, true, false, 4>(p, lds, st);
    else if (big) launch_k1<512, 16, false, false, 4, false, true>(p, lds, st);
    else launch_k1<512, 16, false, false, 4>(p, lds, st);
}

void launch_vmap(const u64* Vraw, u64* V, u64 L, u64 lw, u64 nwords, hipStream_t st)
{
    hipLaunchKernelGGL(k_vmap, dim3(1024), dim3(256), 0, st, Vraw, V, L, lw, nwords);
}
void launch_runstart_words(const u64* V, u64* H, u64* cnt, u64 L, u64 nwords, hipStream_t st)
{
    hipLaunchKernelGGL(k_runstart_words, dim3(1024), dim3(256), 0, st, V, H, cnt, L, nwords);
}
void launch_write_positions(const u64* H, const u64* wbase, u64* pos, u64 nwords, const u64* total, u64 L, hipStream_t st)
{
    hipLaunchKernelGGL(k_write_positions, dim3(1024), dim3(256), 0, st, H, wbase, pos, nwords, total, L);
}
void launch_seg_flags(const u64* run_start, const u64* V, const u64* R_ptr, u64 l, u64* flag, hipStream_t st)
{
    hipLaunchKernelGGL(k_seg_flags, dim3(1024), dim3(256), 0, st, run_start, V, R_ptr, l, flag);
}
void launch_write_segs(const u64* run_start, const u64* flag, const u64* sidx, const u64* R_ptr, const u64* nseg_ptr, u64* seg_start,
                       u64* Hseg, u64 L, hipStream_t st)
{
    hipLaunchKernelGGL(k_write_segs, dim3(1024), dim3(256), 0, st, run_start, flag, sidx, R_ptr, nseg_ptr, seg_start, Hseg, L);
}
void launch_popc_words(const u64* H, u64* cnt, u64 nwords, hipStream_t st)
{
    hipLaunchKernelGGL(k_popc_words, dim3(1024), dim3(256), 0, st, H, cnt, nwords);
}

} // namespace edsx
