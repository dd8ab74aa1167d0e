// msa_scan_launch.hpp - the host side's view of the row index, column scan and runs -> segments kernels.
// They live in a translation unit of their own (msa_scan.hip: compiled with the machine scheduler off, which keeps
// the column scan's loads in source order and measured 26.5 instead of 26.9 ms on the bench shape; the same flag costs
// the emitters 0.2 ms, hence not for the other unit); msa_device.hip launches them through these functions.
#pragma once
#include "msa_device.hpp"

namespace edsx {

// ---------------------------------------------------------------------------------------------
// K1: column scan + variant-column extraction.  One workgroup owns a tile of W = 16*CPR raw
// columns for ALL rows: T threads, thread (sub, j) holds the 16-byte chunk j of rows
// sub, sub+RI, ... (RI = T/CPR) in registers, so each input byte is read from HBM once.
//   msa_transforms.cpp:71-79  B[i] = 0 if c != ref[i] || c == '-'
// ---------------------------------------------------------------------------------------------
struct K1Params {
    const uint8_t* file; const u64* row_start; MsaHdr* hdr;
    u64* Vraw; u64* word_slot; uint8_t* vc; u64 vc_cap_cols;
    u64 Draw, lw; u32 S, Spad, cpr_log2, cap_cols /* LDS colbuf capacity in columns */;
    u64 ntiles;
    // fused grouping (context length 0, one-line rows, S <= 1024): the variant runs that lie inside a tile are
    // grouped right here, from the LDS image of the tile's variant columns; only the other columns go to vc
    u32 fuse; u64* Fraw; u32* rec_info; uint8_t* recf; u32 recf_stride, recf_gid;
};
// fused record (indexed by the vc slot of the run's first column): group ids, 2 bits each (dword l = rows 16l..16l+15)
// for up to 4 strings, 4 bits each (two dwords per lane) for 5..16; then at recf_gid: u32 k | textlen << 8, then the
// .eds text "{s0,s1,..}" (<= REC_TEXT_MAX bytes).  rec_info[slot] = k | textlen << 8 | 4-bit ids << 30 | ok << 31
constexpr u32 REC_TEXT_MAX = 64;

// K0: row index
void launch_find_hdr_end(const uint8_t* f, u64 n, MsaHdr* h, hipStream_t st);
void launch_find_row0(const uint8_t* f, u64 n, MsaHdr* h, hipStream_t st);
void launch_index_spec(const uint8_t* f, u64 n, const MsaHdr* h, u64* hpos, u64* cand, u64 row_cap, hipStream_t st);
void launch_index_check(const uint8_t* f, u64 n, MsaHdr* h, const u64* hpos, const u64* cand, u64* row_start, u64 row_cap, hipStream_t st);
void launch_index_rows(const uint8_t* f, u64 n, MsaHdr* h, u64* row_start, u64 row_cap, hipStream_t st);
void launch_pad_rows(u64* row_start, u64 S, u64 n, hipStream_t st);
// K1: threads per workgroup 512 or 1024; hold: the tile's rows fit the registers; lane_rows: a thread's rows are 16
// consecutive ones; rows64: one row per lane in the fused grouping (up to 64 rows); big: more rows than the LDS holds
void launch_scan_extract(const K1Params& p, int threads, bool hold, bool lane_rows, bool rows64, bool big, size_t lds, hipStream_t st);
// K2: runs -> segments
void launch_vmap(const u64* Vraw, u64* V, u64 L, u64 lw, u64 nwords, hipStream_t st);
void launch_runstart_words(const u64* V, u64* H, u64* cnt, u64 L, u64 nwords, hipStream_t st);
void launch_write_positions(const u64* H, const u64* wbase, u64* pos, u64 nwords, const u64* total, u64 L, hipStream_t st);
void launch_seg_flags(const u64* run_start, const u64* V, const u64* R_ptr, u64 l, u64* flag, hipStream_t st);
void launch_write_segs(const u64* run_start, const u64* flag, const u64* sidx, const u64* R_ptr, const u64* nseg_ptr, u64* seg_start,
                       u64* Hseg, u64 L, hipStream_t st);
void launch_popc_words(const u64* H, u64* cnt, u64 nwords, hipStream_t st);

} // namespace edsx
