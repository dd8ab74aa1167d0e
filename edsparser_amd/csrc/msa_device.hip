// msa_device.hip — MSA -> EDS / l-EDS on gfx950 (MI355X).  Hand-written HIP, wave64, no MFMA:
// this is byte/bit work bounded by HBM bandwidth.
//
// Replaces the reference's three passes (src/cpp/lib/transforms/msa_transforms.cpp):
//   pass 1 parse_msa_and_build_variant_bv :36-90   -> k_find_*, k_index_spec/_check/_rows + k_scan_extract
//   pass 2 build_eds/leds_boundaries      :101-190 -> k_runstart_words .. k_write_segs
//   pass 3 generate_output                :200-324 -> k_seg_meta, k_seg_count_fast (wave per segment, S <= 1024)
//                                                     / k_seg_count (workgroup per segment), scans,
//                                                     k_emit_fast2 / k_emit_fast / k_emit_variant (+ the common text in front)
//
// Data layout in HBM
//   file image    the FASTA bytes as given (no repacking).  Row r's raw byte q lives at
//                 row_start[r] + q; raw position q of alignment column c is c + c/lw for wrapped
//                 rows (msa_transforms.cpp:268-269) and c for one-line rows.
//   Vraw / V      1 bit per raw position / alignment column: 1 = variant column (some row differs
//                 from row 0, or row 0 has '-').  Complement of the reference's bit-vector B.
//   vc            the variant columns only, column-major: vc[slot * Spad + r] = byte of row r
//                 (natural row order: a lane's 16-byte load at 16*lane holds rows 16*lane .. +15).
//                 Slots are handed out per tile by an atomic counter (unordered between tiles,
//                 consecutive inside a 64-column word); word_slot[w] = slot of the first variant
//                 column of raw word w.  Every later kernel reads rows through vc, so the S x L
//                 matrix is read from HBM exactly once.
//   run/seg table seg_start[nseg+1] (alignment columns), bitmap Hseg of segment starts with a
//                 per-word prefix segbase[]; a segment is common iff V[seg_start] == 0.
//   rec           one grouping record per variant segment (count -> emit): group id of every row (2 or 4
//                 bits, natural row order), number of strings, representative rows / the .eds text.
//   eds_off/seds_off  exclusive scans of the per-segment text sizes.
//
// Source layout: this file is the host side (MsaPipeline: sizing, launches, timers) and the translation unit of the
// grouping / text kernels; the device code is in msa_wave.hpp (wave64 helpers, grouping primitives), msa_scan_kernels.hpp
// (row index, column scan, runs -> segments: a translation unit of their own, msa_scan.hip, launched through
// msa_scan_launch.hpp), msa_generic_kernels.hpp (workgroup per segment, common text) and
// msa_fast_kernels.hpp (wave per segment, up to 1024 rows: tables, grouping, emitters) and msa_rowloop_kernels.hpp (wave
// per segment, any number of rows).
#include "msa_device.hpp"
#include "msa_wave.hpp"
#include "msa_scan_launch.hpp"
#include "msa_generic_kernels.hpp"
#include "msa_fast_kernels.hpp"
#include "msa_rowloop_kernels.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace edsx {

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static u64 token_total(u32 S)
{
    u64 t = 0;
    for (u32 r = 1; r <= S; r++) { u32 d = 1; for (u32 v = r; v >= 10; v /= 10) d++; t += d + 1; }
    return t;
}

void MsaPipeline::launch_timer_begin(const char* name, hipStream_t st)
{
    if (!timing_) return;
    TimedKernel tk; tk.name = name;
    EDSX_HIP(hipEventCreate(&tk.t0)); EDSX_HIP(hipEventCreate(&tk.t1));
    EDSX_HIP(hipEventRecord(tk.t0, st));
    timed_.push_back(tk);
}
void MsaPipeline::launch_timer_end(hipStream_t st)
{
    if (!timing_) return;
    EDSX_HIP(hipEventRecord(timed_.back().t1, st));
}
// harvest the finished event pairs of the previous call into the per-kernel accumulators
void MsaPipeline::clear_timers()
{
    for (auto& t : timed_) {
        float v = 0;
        (void)hipEventSynchronize(t.t1);
        if (hipEventElapsedTime(&v, t.t0, t.t1) == hipSuccess) {
            bool found = false;
            for (auto& a : acc_) if (a.name == t.name) { a.total_ms += v; a.count++; found = true; break; }
            if (!found) acc_.push_back({t.name, v, 1});
        }
        (void)hipEventDestroy(t.t0); (void)hipEventDestroy(t.t1);
    }
    timed_.clear();
}
void MsaPipeline::set_timing(bool on)
{
    clear_timers();
    acc_.clear();
    timing_ = on;
}
// accumulated device time per kernel since set_timing(true): ms[i] = total, counts[i] = launches
int MsaPipeline::get_timing(const char** names, float* ms, int* counts, int cap)
{
    clear_timers();
    int n = 0;
    for (const auto& a : acc_) {
        if (n >= cap) break;
        names[n] = a.name; ms[n] = (float)a.total_ms; counts[n] = a.count; n++;
    }
    return n;
}

#define TIMED(name, st, ...) do { launch_timer_begin(name, st); __VA_ARGS__; launch_timer_end(st); } while (0)

MsaPipeline::~MsaPipeline()
{
    clear_timers();
    if (side_ready_) {
        for (auto& x : side_) (void)hipStreamDestroy(x);
        for (auto& e : side_ev_) (void)hipEventDestroy(e);
    }
}

void MsaPipeline::ensure_side_streams()
{
    if (side_ready_) return;
    // lowest priority: where they run beside the main emitter (EDSX_SIDE=2) they only take what it leaves free
    int least = 0, greatest = 0;
    EDSX_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    for (auto& x : side_) EDSX_HIP(hipStreamCreateWithPriority(&x, hipStreamNonBlocking, least));
    for (auto& e : side_ev_) EDSX_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    side_ready_ = true;
}

static const char* status_message(u64 st)
{
    if (st & ST_NOT_FASTA) return "Invalid MSA: expected a FASTA header line starting with '>'";
    if (st & ST_FEW_ROWS) return "Invalid MSA: at least two sequences are required";
    if (st & ST_TOO_MANY_ROWS) return "MSA has more sequences than this build supports (9999999)";
    if (st & ST_NEWLINE_IN_DATA) return "Invalid MSA: rows must have equal length and a uniform line width";
    if (st & ST_LAYOUT) return "Invalid MSA: rows must have equal length and a uniform line width";
    return "MSA transform failed";
}

void MsaPipeline::plan(const uint8_t* d_msa, size_t n, uint32_t l, hipStream_t st,
                       uint64_t* eds_bytes, uint64_t* seds_bytes)
{
    clear_timers();
    planned_ = false;
    file_ = d_msa; n_ = n; l_ = l;
    if (n == 0) throw FormatError("Invalid MSA: empty input");

    hdr_.ensure(sizeof(MsaHdr));
    MsaHdr* dh = hdr_.as<MsaHdr>();
    // ---- K0: geometry + row index (one host sync: everything below is sized from it).  The row table starts with room
    // for ROW_CAP rows; an alignment with more rows is indexed again with the table sized from its first row's length.
    u64 row_cap = ROW_CAP;
    for (int attempt = 0;; attempt++) {
        rows_.ensure(sizeof(u64) * (row_cap + ROW_PAD));
        idx_tmp_.ensure(2 * sizeof(u64) * row_cap);
        EDSX_HIP(hipMemsetAsync(dh, 0, sizeof(MsaHdr), st));
        TIMED("k_find_hdr_end", st, launch_find_hdr_end(d_msa, (u64)n, dh, st));
        TIMED("k_find_row0", st, launch_find_row0(d_msa, (u64)n, dh, st));
        TIMED("k_index_spec", st, launch_index_spec(d_msa, (u64)n, dh, idx_tmp_.as<u64>(), idx_tmp_.as<u64>() + row_cap, row_cap, st));
        TIMED("k_index_check", st, launch_index_check(d_msa, (u64)n, dh, idx_tmp_.as<u64>(), idx_tmp_.as<u64>() + row_cap, rows_.as<u64>(), row_cap, st));
        TIMED("k_index_rows", st, launch_index_rows(d_msa, (u64)n, dh, rows_.as<u64>(), row_cap, st));
        EDSX_HIP(hipMemcpyAsync(&h_, dh, sizeof(MsaHdr), hipMemcpyDeviceToHost, st));
        EDSX_HIP(hipStreamSynchronize(st));
        if (attempt == 0 && h_.status == ST_TOO_MANY_ROWS) {
            const u64 most = n / (h_.Draw + 3) + 2;            // ">\n" + the row + "\n" at least
            if (most > row_cap && row_cap <= MAX_ROWS) { row_cap = std::min<u64>(most, MAX_ROWS + 1); clear_timers(); continue; }
        }
        break;
    }
    if (h_.status & ST_TOO_MANY_ROWS) throw LimitError(status_message(ST_TOO_MANY_ROWS));
    if (h_.status) throw FormatError(status_message(h_.status));
    if (h_.S > MAX_ROWS) throw LimitError(status_message(ST_TOO_MANY_ROWS));
    {   // padding for the column scan's 16-row loads (at most 256 threads x 16 rows past the last one)
        const u64 padded = std::min<u64>(row_cap + ROW_PAD, (h_.S + 15) / 16 * 16 + 4096);
        launch_pad_rows(rows_.as<u64>(), h_.S, padded, st);
    }

    const u64 Draw = h_.Draw;
    const u32 Spad = vc_pitch((u32)h_.S);
    {   // first guess for vc: 12.5 % variant columns; grown to the exact need on overflow
        u64 want = std::max<u64>(Draw / 8, 4096);
        if (want > Draw) want = Draw;
        vc_.ensure((size_t)want * Spad);
    }
    for (int attempt = 0;; attempt++) {
        vc_cap_cols_ = vc_.cap / Spad;
        plan_body(st);
        EDSX_HIP(hipMemcpyAsync(&h_, dh, sizeof(MsaHdr), hipMemcpyDeviceToHost, st));
        EDSX_HIP(hipStreamSynchronize(st));
        EDSX_HIP(hipGetLastError());
        if ((h_.status & ST_VC_OVERFLOW) && !(h_.status & ~(u64)ST_VC_OVERFLOW) && attempt == 0) {
            vc_.ensure((size_t)h_.nv * Spad);            // K1 counted every variant column
            EDSX_HIP(hipMemsetAsync(&dh->nv, 0, sizeof(u64), st));
            EDSX_HIP(hipMemsetAsync(&dh->status, 0, sizeof(u64), st));
            clear_timers();
            continue;
        }
        break;
    }
    if (h_.status) throw FormatError(status_message(h_.status & ~(u64)ST_VC_OVERFLOW));
    if (getenv("EDSX_COUNTS"))
        fprintf(stderr, "[edsx] segments %llu variant %llu | grouping list %llu heavy %llu | wide emit %llu | generic %llu + %llu | fused %d\n",
                (unsigned long long)(l == 0 ? h_.R : h_.nseg), (unsigned long long)h_.nvs, (unsigned long long)h_.cnt_n,
                (unsigned long long)h_.heavy_n, (unsigned long long)(h_.wide_n + h_.wide16_n), (unsigned long long)h_.slow_n,
                (unsigned long long)h_.slow_n2, (int)fuse_);
    if (l == 0) h_.nseg = h_.R;
    *eds_bytes = h_.E;
    *seds_bytes = h_.Q;
    planned_ = true;
}

// everything between the row index and the output sizes; no host synchronisation inside
void MsaPipeline::plan_body(hipStream_t st)
{
    MsaHdr* dh = hdr_.as<MsaHdr>();
    const uint8_t* d_msa = file_;
    const uint32_t l = l_;
    const u64 S = h_.S, L = h_.L, lw = h_.lw, Draw = h_.Draw;
    const u32 Spad = vc_pitch((u32)S);
    const u64 nwords = h_.nwords, nwords_raw = h_.nwords_raw;

    // ---- workspace (sized by the worst case "every column starts a run")
    vraw_.ensure(8 * (nwords_raw + 1));
    wslot_.ensure(8 * (nwords_raw + 1));
    if (lw) v_.ensure(8 * (nwords + 1));
    hrun_.ensure(8 * (nwords + 1));
    hseg_.ensure(8 * (nwords + 1));
    cnt_.ensure(8 * (nwords + 1));
    wbase_.ensure(8 * (nwords + 1));
    segbase_.ensure(8 * (nwords + 1));
    scan_tmp_.ensure(8 * 4 * ((L + 2) / SCAN_TILE + 4));      // up to four arrays per pass (exclusive_scan_multi)
    run_start_.ensure(8 * (L + 2));
    if (l) { flag_.ensure(8 * (L + 2)); seg_start_.ensure(8 * (L + 2)); }
    eds_len_.ensure(8 * (L + 2));
    seds_len_.ensure(8 * (L + 2));

    // ---- K1 geometry.  A workgroup of T threads keeps RPT 16-byte chunks per thread in registers:
    // rows per pass = T / CPR, tile width W = 16 * CPR columns.  Two workgroups per CU let one
    // tile's extraction phase overlap the other's load phase.
    // T = 512, RPT = 16: two workgroups per CU (1024 threads x 16 and 1024 x 8 were measured in round 1: slower).
    // 2049 .. 4096 rows: 1024 threads (one workgroup per CU) still hold 64-column tiles of all rows in registers; beyond
    // that the rows are walked in a loop (twice: masks, then the variant bytes).
    const int T = (S > 2048 && S <= 4096) ? 1024 : 512, RPT = 16;
    u32 cpr_log2 = 8;
    while (cpr_log2 > 2 && (u64)RPT * (T >> cpr_log2) < S) cpr_log2--;
    const bool hold = (u64)RPT * (T >> cpr_log2) >= S;
    // (128-byte instead of 64-byte row pieces for the rows that do not fit the registers - cpr_log2 = 3 when !hold - were
    // measured in round 3: 3000 rows x 1 M columns, scan 3.8 -> 6.0 ms: twice the threads re-read their rows for the columns)
    const u64 W = 16ull << cpr_log2;
    const u64 ntiles = (Draw + W - 1) / W;
    // 40 KB of column image (39 columns of 1000 rows; denser tiles take the batched path) leave room for a third
    // workgroup per CU to start while the two resident ones finish their grouping tails (-1.7 % on the scan)
    big_ = S > LDS_ROWS;                                  // row tables / column image in HBM instead of LDS
    const size_t colbuf_bytes = big_ ? 64 : (size_t)S * 8 <= 40 * 1024 ? 40 * 1024 : 64 * 1024;
    if (ntiles > 0x7fffffffull) throw FormatError("MSA too large for one launch");

    K1Params kp;
    kp.file = d_msa; kp.row_start = rows_.as<u64>(); kp.hdr = dh;
    kp.Vraw = vraw_.as<u64>(); kp.word_slot = wslot_.as<u64>(); kp.vc = vc_.as<uint8_t>();
    kp.vc_cap_cols = vc_cap_cols_; kp.Draw = Draw; kp.lw = lw; kp.S = (u32)S; kp.Spad = Spad;
    kp.cpr_log2 = cpr_log2; kp.cap_cols = (u32)(colbuf_bytes / Spad); kp.ntiles = ntiles;
    const bool lane_rows = hold && RPT == 16;             // thread rows = 16 consecutive rows = 16 consecutive vc bytes
    // fused grouping: context length 0 (segments = runs), one-line rows (raw position = column), wave-per-segment code
    fuse_ = l == 0 && lw == 0 && S <= 1024 && lane_rows;
    kp.fuse = fuse_ ? 1u : 0u; kp.Fraw = nullptr; kp.rec_info = nullptr; kp.recf = nullptr; kp.recf_stride = 0; kp.recf_gid = 0;
    if (fuse_) {
        recf_gid_ = (8u * (((u32)S + 15u) / 16u) + 15u) & ~15u;
        recf_stride_ = (recf_gid_ + 4u + REC_TEXT_MAX + 63u) & ~63u;
        fraw_.ensure(8 * (nwords_raw + 1));
        rec_info_.ensure(4 * ((size_t)vc_cap_cols_ + 2));
        recf_.ensure(((size_t)vc_cap_cols_ + 2) * recf_stride_);
        kp.Fraw = fraw_.as<u64>(); kp.rec_info = rec_info_.as<u32>(); kp.recf = recf_.as<uint8_t>();
        kp.recf_stride = recf_stride_; kp.recf_gid = recf_gid_;
    }
    launch_timer_begin("k_scan_extract", st);
    launch_scan_extract(kp, T, hold, lane_rows, /*rows64=*/lane_rows && fuse_ && S <= 64, big_, colbuf_bytes, st);
    launch_timer_end(st);

    const u64* V = lw ? v_.as<u64>() : vraw_.as<u64>();
    if (lw) TIMED("k_vmap", st, launch_vmap(vraw_.as<u64>(), v_.as<u64>(), L, lw, nwords, st));

    // ---- K2: runs -> segments
    u64* d_nwords = &dh->nwords;
    TIMED("k_runstart_words", st, launch_runstart_words(V, hrun_.as<u64>(), cnt_.as<u64>(), L, nwords, st));
    TIMED("scan_runs", st, exclusive_scan_u64(cnt_.as<u64>(), wbase_.as<u64>(), d_nwords, &dh->R,
                                              scan_tmp_.as<u64>(), st));
    TIMED("k_write_runs", st, launch_write_positions(hrun_.as<u64>(), wbase_.as<u64>(), run_start_.as<u64>(), nwords, &dh->R, L, st));
    const u64* seg_start; const u64* Hseg; const u64* segbase; const u64* d_nseg;
    if (l == 0) {
        seg_start = run_start_.as<u64>(); Hseg = hrun_.as<u64>(); segbase = wbase_.as<u64>(); d_nseg = &dh->R;
    } else {
        TIMED("k_seg_flags", st, launch_seg_flags(run_start_.as<u64>(), V, &dh->R, (u64)l, flag_.as<u64>(), st));
        // eds_len_ doubles as scratch for the run-sized scan (it is overwritten by k_seg_count later)
        TIMED("scan_flags", st, exclusive_scan_u64(flag_.as<u64>(), eds_len_.as<u64>(), &dh->R, &dh->nseg,
                                                   scan_tmp_.as<u64>(), st));
        EDSX_HIP(hipMemsetAsync(hseg_.ptr, 0, 8 * (nwords + 1), st));
        TIMED("k_write_segs", st, launch_write_segs(run_start_.as<u64>(), flag_.as<u64>(), eds_len_.as<u64>(), &dh->R, &dh->nseg,
                                                    seg_start_.as<u64>(), hseg_.as<u64>(), L, st));
        TIMED("k_popc_words", st, launch_popc_words(hseg_.as<u64>(), cnt_.as<u64>(), nwords, st));
        TIMED("scan_segwords", st, exclusive_scan_u64(cnt_.as<u64>(), segbase_.as<u64>(), d_nwords, &dh->tmp_total,
                                                      scan_tmp_.as<u64>(), st));
        seg_start = seg_start_.as<u64>(); Hseg = hseg_.as<u64>(); segbase = segbase_.as<u64>(); d_nseg = &dh->nseg;
    }

    // ---- K3 + K4: per-segment sizes, offsets
    mv_.file = d_msa; mv_.row_start = rows_.as<u64>(); mv_.V = V; mv_.Vraw = vraw_.as<u64>();
    mv_.word_slot = wslot_.as<u64>(); mv_.vc = vc_.as<uint8_t>(); mv_.hdr = dh; mv_.L = L; mv_.lw = lw;
    mv_.S = (u32)S; mv_.Spad = Spad; mv_.tileW = (u32)W;
    seg_start_p_ = seg_start; hseg_p_ = Hseg; segbase_p_ = segbase; nseg_p_ = d_nseg;
    seg_lds_ = (size_t)18 * S + 64 + (S <= HT_MAX_ROWS ? HtLds::bytes(ht_size_of((u32)S)) + 16 : 0);
    seg_lds_ = (seg_lds_ + 15) & ~(size_t)15;
    stage_off_ = 0;
    stage_cols_ = 0;
    seg_scratch_stride_ = 0;
    if (big_) {
        // the generic kernels' row tables (20 B per row) and the emitter's walk tables: a slice of HBM per workgroup;
        // as many workgroups as 8 GiB of it allow (32 .. 512)
        seg_lds_ = 64;
        seg_scratch_stride_ = (((SegLdsT<true>::bytes((u32)S) + 15) & ~(size_t)15) + walk_table_bytes<true>((u32)S) + 255) & ~(size_t)255;
        big_grid_ = (unsigned)std::max<u64>(32, std::min<u64>(512, (8ull << 30) / seg_scratch_stride_));
        seg_scratch_.ensure((size_t)big_grid_ * seg_scratch_stride_);
    } else {                                                     // as many columns of a segment as fit beside that (<= STAGE_COLS)
        const size_t budget = (size_t)150 * 1024;
        const size_t maps = 2 * (size_t)STAGE_WMAX + 16;     // column map + reference bytes of the common columns
        constexpr size_t GEN_MINCOLS = 24;
        // two (or three) workgroups per CU if 24 columns and more still fit - 8 for more than 1024 rows, which have no other
        // path (wider segments keep their variant columns + a column map)
        const size_t nblk = (S + 63) / 64, S2 = (S + 1) & ~(size_t)1;
        const size_t walk_cols = (nblk * 384 + S2 * 3 + nblk + 16 + Spad + 7) / ((size_t)Spad + 8);     // the emitter's tables for the .seds walk live there too
        const size_t mincols = S > 1024 ? std::max<size_t>(8, walk_cols) : GEN_MINCOLS;
        size_t use = budget;
        {
            const size_t part = (size_t)(150 / 2 - 1) * 1024;
            if (seg_lds_ + maps + mincols * ((size_t)Spad + 8) <= part) use = part;
        }
        if (seg_lds_ + maps + 4 * ((size_t)Spad + 8) <= use) {
            stage_cols_ = (u32)std::min<size_t>(STAGE_COLS, (use - seg_lds_ - maps) / ((size_t)Spad + 8));
            stage_off_ = (u32)seg_lds_;
            seg_lds_ += stage_cols_offset(stage_cols_) + (size_t)stage_cols_ * Spad;
        }
    }

    // grouping cache of the generic kernels (count -> emit): two regions (one per work list; without lists: both)
    gc_stride_ = gcache_stride_of((u32)S, big_);
    // (with the wave-per-segment kernels in front - up to 1024 rows - only the few segments on their slow lists come here:
    // 256 MiB per list; items beyond the cache are simply grouped again by the emitter)
    gc_region_ = (size_t)std::min<u64>(std::max<u64>((u64)(S <= 1024 ? 256u : 1024u) << 20, 4 * gc_stride_), (L / 2 + 4) * (u64)gc_stride_);
    gc_region_ = gc_region_ / gc_stride_ * gc_stride_;
    gcache_.ensure(2 * gc_region_ + 16);

    const u64 tok_total = token_total((u32)S);
    fast_ = S <= 1024;
    SegParams sp;
    sp.mv = mv_; sp.seg_start = seg_start; sp.nseg_ptr = d_nseg; sp.eds_len = eds_len_.as<u64>();
    sp.seds_len = seds_len_.as<u64>(); sp.tok_total = tok_total; sp.list = nullptr; sp.list_n = nullptr;
    sp.stage_cols = stage_cols_; sp.stage_off = stage_off_;
    long_list_.ensure(8 * (L / LONG_COMMON + 4));             // the long common segments
    sp.long_list = long_list_.as<u64>(); sp.long_count = &dh->long_n;
    EDSX_HIP(hipMemsetAsync(&dh->long_n, 0, sizeof(u64), st));
    sp.scratch = seg_scratch_.as<uint8_t>(); sp.scratch_stride = seg_scratch_stride_;
    EDSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_seg_count<false>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)seg_lds_));
    if (fast_) {
        segmeta_.ensure(8 * (L + 2));
        // list 1: too wide or mixed segments (k_seg_meta), list 2: those the grouping kernel gives up on; together
        // at most all variant segments (<= L/2 + 1)
        slow_list_.ensure(8 * 4 * (L / 2 + 4));            // + the wide emitters' two lists
        cnt_list_.ensure(8 * 11 * (L / 2 + 4));               // work lists of the grouping kernels (ordinal + column descriptor, three times), descriptor and flags per variant segment
        fp_.mv = mv_; fp_.seg_start = seg_start; fp_.nseg_ptr = d_nseg; fp_.segmeta = segmeta_.as<u64>();
        fp_.eds_len = eds_len_.as<u64>(); fp_.seds_len = seds_len_.as<u64>();
        fp_.slow_list = slow_list_.as<u64>(); fp_.slow_count = &dh->slow_n;
        fp_.slow_list2 = slow_list_.as<u64>() + (L / 2 + 4); fp_.slow_count2 = &dh->slow_n2;
        fp_.cnt_vi = cnt_list_.as<u64>(); fp_.cnt_cm = fp_.cnt_vi + (L / 2 + 4); fp_.cnt_n = &dh->cnt_n;
        fp_.heavy_vi = fp_.cnt_cm + (L / 2 + 4); fp_.heavy_cm = fp_.heavy_vi + (L / 2 + 4); fp_.heavy_n = &dh->heavy_n;
        fp_.cnt_meta = fp_.heavy_cm + (L / 2 + 4); fp_.cnt_flag = fp_.cnt_meta + (L / 2 + 4); fp_.wide_flag = fp_.cnt_flag + (L / 2 + 4);
        fp_.heavy_flag = fp_.wide_flag + (L / 2 + 4); fp_.heavy2_vi = fp_.heavy_flag + (L / 2 + 4); fp_.heavy2_cm = fp_.heavy2_vi + (L / 2 + 4);
        fp_.heavy2_n = &dh->heavy2_n;
        fp_.wide16_flag = fp_.heavy2_cm + (L / 2 + 4);
        fp_.wide_list = slow_list_.as<u64>() + 2 * (L / 2 + 4); fp_.wide_count = &dh->wide_n;
        fp_.wide16_list = slow_list_.as<u64>() + 3 * (L / 2 + 4); fp_.wide16_count = &dh->wide16_n;
        fp_.eds = nullptr; fp_.seds = nullptr; fp_.tok_total = tok_total;
        // one record per variant segment; there are at most as many as variant columns
        fp_.rec_stride = rec_stride((u32)S); fp_.rec_gid = rec_gid_bytes((u32)S);
        rec_.ensure(((size_t)std::min<u64>(vc_cap_cols_, L / 2 + 2) + 2) * fp_.rec_stride);
        fp_.rec = rec_.as<uint8_t>();
        fp_.Fraw = fuse_ ? fraw_.as<u64>() : nullptr; fp_.rec_info = rec_info_.as<u32>();
        fp_.recf = recf_.as<uint8_t>(); fp_.recf_stride = recf_stride_; fp_.recf_gid = recf_gid_;
        fp_.long_list = sp.long_list; fp_.long_count = sp.long_count;
        EDSX_HIP(hipMemsetAsync(&dh->slow_n, 0, 2 * sizeof(u64), st));
        EDSX_HIP(hipMemsetAsync(&dh->wide_n, 0, 3 * sizeof(u64), st));     // (wide_n is then set by scan_wide, and added to by k_seg_group)
        EDSX_HIP(hipMemsetAsync(&dh->heavy2_n, 0, 2 * sizeof(u64), st));    // + wide16_n
        TIMED("k_seg_meta", st, hipLaunchKernelGGL(k_seg_meta, dim3(4096), dim3(256), 0, st, fp_));
        {   // list positions of the light / heavy grouping lists and of the wide emitter's list: one pass
            ScanSet<4> ss{{fp_.cnt_flag, fp_.heavy_flag, fp_.wide_flag, fp_.wide16_flag}, {fp_.cnt_flag, fp_.heavy_flag, fp_.wide_flag, fp_.wide16_flag},
                          {&dh->cnt_n, &dh->heavy_n, &dh->wide_n, &dh->wide16_n}};
            TIMED("scan_lists", st, exclusive_scan_multi<4>(ss, &dh->nvs, scan_tmp_.as<u64>(), st));
        }
        TIMED("k_work_scatter", st, hipLaunchKernelGGL(k_work_scatter, dim3(2048), dim3(256), 0, st, fp_, fp_.cnt_flag, fp_.heavy_flag, &dh->nvs));
        // (The light and the heavy grouping kernel and the generic count are independent, but side by side on three
        // streams they only share the machine - measured in round 3: 1.85 ms for the three against 1.33 + 0.41 + 0.01 one
        // after the other - so they run in line.)
        TIMED("k_seg_group", st, hipLaunchKernelGGL(k_seg_group<false>, dim3(persistent_grid(
                  reinterpret_cast<const void*>(k_seg_group<false>), 256, 0)), dim3(256), 0, st, fp_, fp_.cnt_vi, fp_.cnt_cm, &dh->cnt_n));
        TIMED("k_seg_group_heavy", st, hipLaunchKernelGGL(k_seg_group<true>, dim3(persistent_grid(
                  reinterpret_cast<const void*>(k_seg_group<true>), 256, 0)), dim3(256), 0, st, fp_, fp_.heavy_vi, fp_.heavy_cm, &dh->heavy_n));
        // second heavy pass: what the light kernel could not group (alphabets other than {A,C,G,T,N,-}; usually nothing)
        TIMED("k_seg_group_heavy2", st, hipLaunchKernelGGL(k_seg_group<true>, dim3(persistent_grid(
                  reinterpret_cast<const void*>(k_seg_group<true>), 256, 0)), dim3(256), 0, st, fp_, fp_.heavy2_vi, fp_.heavy2_cm, &dh->heavy2_n));
        sp.gcache_stride = gc_stride_; sp.gcache_cap = gc_region_ / gc_stride_;
        sp.list = fp_.slow_list; sp.list_n = fp_.slow_count; sp.gcache = gcache_.as<uint8_t>();
        TIMED("k_seg_count_slow", st, hipLaunchKernelGGL(k_seg_count<false>, dim3(seg_grid()), dim3(GT), seg_lds_, st, sp));
        sp.list = fp_.slow_list2; sp.list_n = fp_.slow_count2; sp.gcache = gcache_.as<uint8_t>() + gc_region_;
        TIMED("k_seg_count_slow2", st, hipLaunchKernelGGL(k_seg_count<false>, dim3(seg_grid()), dim3(GT), seg_lds_, st, sp));
    } else {
        // More than 1024 rows: a wave walks the rows of a variant segment 64 at a time (msa_rowloop_kernels.hpp: up to
        // eight pure variant columns, up to 64 strings - nearly all segments); what it leaves goes to the generic kernels
        // as a work list.
        segmeta_.ensure(8 * (L + 2));
        slow_list_.ensure(8 * (L / 2 + 4));
        rl_.mv = mv_; rl_.seg_start = seg_start; rl_.nseg_ptr = d_nseg; rl_.eds_len = eds_len_.as<u64>(); rl_.seds_len = seds_len_.as<u64>();
        rl_.tok_total = tok_total; rl_.segmeta = segmeta_.as<u64>(); rl_.slow_list = slow_list_.as<u64>(); rl_.slow_count = &dh->slow_n;
        rl_.long_list = sp.long_list; rl_.long_count = sp.long_count;
        rl_.rec_stride = rl_stride_of((u32)S);
        rec_.ensure(((size_t)std::min<u64>(vc_cap_cols_, L / 2 + 2) + 2) * rl_.rec_stride);   // a record per variant segment (<= variant columns)
        rl_.rec = rec_.as<uint8_t>(); rl_.eds = nullptr; rl_.seds = nullptr; rl_.next = &dh->rl_next;
        EDSX_HIP(hipMemsetAsync(&dh->slow_n, 0, 2 * sizeof(u64), st));
        EDSX_HIP(hipMemsetAsync(&dh->rl_next, 0, sizeof(u64), st));
        TIMED("k_rl_count", st, hipLaunchKernelGGL(k_rl_count, dim3(persistent_grid(reinterpret_cast<const void*>(k_rl_count), 256, 0)),
                                                   dim3(256), 0, st, rl_));
        sp.gcache_stride = gc_stride_; sp.gcache_cap = 2 * gc_region_ / gc_stride_; sp.gcache = gcache_.as<uint8_t>();
        sp.list = rl_.slow_list; sp.list_n = rl_.slow_count;
        if (big_) TIMED("k_seg_count", st, hipLaunchKernelGGL(k_seg_count<true>, dim3(seg_grid()), dim3(GT), seg_lds_, st, sp));
        else TIMED("k_seg_count", st, hipLaunchKernelGGL(k_seg_count<false>, dim3(seg_grid()), dim3(GT), seg_lds_, st, sp));
    }
    {   // text offsets of both outputs: one pass
        ScanSet<2> ss{{eds_len_.as<u64>(), seds_len_.as<u64>()}, {eds_len_.as<u64>(), seds_len_.as<u64>()}, {&dh->E, &dh->Q}};
        TIMED("scan_text", st, exclusive_scan_multi<2>(ss, d_nseg, scan_tmp_.as<u64>(), st));
    }
}

// workgroups that are resident at once: one persistent workgroup per slot
unsigned MsaPipeline::persistent_grid(const void* kern, int threads, size_t dyn_lds) const
{
    int per_cu = 1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, threads, dyn_lds) != hipSuccess || per_cu < 1)
        per_cu = 1;
    if (!cus_) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        const_cast<MsaPipeline*>(this)->cus_ = cus;
    }
    return (unsigned)(cus_ * per_cu);
}

unsigned MsaPipeline::seg_grid() const
{
    // persistent workgroups striding over the segments; several per CU to hide latency
    if (big_) return big_grid_;
    size_t per_cu = std::max<size_t>(1, std::min<size_t>(8, (150 * 1024) / std::max<size_t>(seg_lds_, 1)));
    return (unsigned)(256 * per_cu);
}

__global__ void k_copy_columns(MsaView mv, u64 col0, u64 ncols, uint8_t* __restrict__ out)
{
    const u64 total = (u64)mv.S * ncols;
    for (u64 t = blockIdx.x * (u64)blockDim.x + threadIdx.x; t < total; t += (u64)gridDim.x * blockDim.x) {
        const u64 r = t / ncols, c = col0 + t % ncols;
        out[t] = mv.file[mv.row_start[r] + mv.raw(c)];
    }
}

// first / last segment of the planned alignment: type, width and text sizes (for the slab stitch).  One small kernel
// gathers the ten numbers, one copy brings them over (the stitch runs inside every multi-GPU step).
__global__ void k_edge_info(const u64* __restrict__ seg_start, const u64* __restrict__ eds_off, const u64* __restrict__ seds_off,
                            const u64* __restrict__ V, u64 nseg, u64 L, u64* __restrict__ out)
{
    if (threadIdx.x || blockIdx.x) return;
    out[0] = seg_start[0]; out[1] = nseg > 1 ? seg_start[1] : L; out[2] = seg_start[nseg - 1];
    out[3] = nseg > 1 ? eds_off[1] : 0; out[4] = nseg > 1 ? seds_off[1] : 0;
    out[5] = eds_off[nseg - 1]; out[6] = seds_off[nseg - 1];
    out[7] = V[0] & 1ull; out[8] = (V[(L - 1) >> 6] >> ((L - 1) & 63)) & 1ull;
}

MsaPipeline::Edges MsaPipeline::edge_info(hipStream_t st)
{
    if (!planned_) throw ParamError("edge_info needs a planned alignment");
    const u64 nseg = h_.nseg;
    Edges e{};
    e.nseg = nseg;
    idx_tmp_.ensure(16 * sizeof(u64));
    u64 h[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    hipLaunchKernelGGL(k_edge_info, dim3(1), dim3(64), 0, st, seg_start_p_, eds_len_.as<u64>(), seds_len_.as<u64>(), mv_.V, nseg, h_.L,
                       idx_tmp_.as<u64>());
    EDSX_HIP(hipMemcpyAsync(h, idx_tmp_.ptr, sizeof(h), hipMemcpyDeviceToHost, st));
    EDSX_HIP(hipStreamSynchronize(st));
    e.fvar = h[7];
    e.fcols = h[1] - h[0];
    e.feds = nseg > 1 ? h[3] : h_.E;
    e.fseds = nseg > 1 ? h[4] : h_.Q;
    e.lvar = h[8];
    e.lcols = h_.L - h[2];
    e.leds = h_.E - h[5];
    e.lseds = h_.Q - h[6];
    return e;
}

// l-EDS stitch: the first and the last standalone common run of at least min_cols columns (msa_transforms.cpp:153) - the
// text between two such anchors does not depend on anything outside them
__global__ void k_anchor_find(const u64* __restrict__ seg_start, const u64* __restrict__ V, u64 nseg, u64 min_cols, u64* __restrict__ out)
{
    for (u64 seg = blockIdx.x * (u64)blockDim.x + threadIdx.x; seg < nseg; seg += (u64)gridDim.x * blockDim.x) {
        const u64 a = seg_start[seg], b = seg_start[seg + 1];
        const bool variant = (V[a >> 6] >> (a & 63)) & 1ull;
        if (!variant && b - a >= min_cols) {
            atomicMin((unsigned long long*)&out[0], (unsigned long long)seg);
            atomicMax((unsigned long long*)&out[1], (unsigned long long)(seg + 1));
        }
    }
}
__global__ void k_anchor_read(const u64* __restrict__ seg_start, const u64* __restrict__ eds_off, const u64* __restrict__ seds_off,
                              u64 nseg, u64 E, u64 Q, u64* __restrict__ out)
{
    if (threadIdx.x || blockIdx.x) return;
    const u64 first = out[0], last1 = out[1];
    if (last1 == 0) return;                                   // none
    const u64 last = last1 - 1;
    out[2] = seg_start[last]; out[3] = eds_off[last]; out[4] = seds_off[last];
    out[5] = seg_start[first + 1];
    out[6] = first + 1 < nseg ? eds_off[first + 1] : E;
    out[7] = first + 1 < nseg ? seds_off[first + 1] : Q;
}

MsaPipeline::Anchors MsaPipeline::anchor_info(u64 min_cols, hipStream_t st)
{
    if (!planned_) throw ParamError("anchor_info needs a planned alignment");
    Anchors a{};
    a.nseg = h_.nseg;
    idx_tmp_.ensure(16 * sizeof(u64));
    u64 h[8] = {~0ull, 0, 0, 0, 0, 0, 0, 0};
    EDSX_HIP(hipMemcpyAsync(idx_tmp_.ptr, h, sizeof(h), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_anchor_find, dim3(1024), dim3(256), 0, st, seg_start_p_, mv_.V, h_.nseg, min_cols, idx_tmp_.as<u64>());
    hipLaunchKernelGGL(k_anchor_read, dim3(1), dim3(64), 0, st, seg_start_p_, eds_len_.as<u64>(), seds_len_.as<u64>(), h_.nseg,
                       h_.E, h_.Q, idx_tmp_.as<u64>());
    EDSX_HIP(hipMemcpyAsync(h, idx_tmp_.ptr, sizeof(h), hipMemcpyDeviceToHost, st));
    EDSX_HIP(hipStreamSynchronize(st));
    if (h[1] == 0) return a;                                  // no such run
    a.found = 1;
    a.first_seg = h[0]; a.last_seg = h[1] - 1;
    a.last_col = h[2]; a.last_eds = h[3]; a.last_seds = h[4];
    a.first_end = h[5]; a.first_eds_end = h[6]; a.first_seds_end = h[7];
    return a;
}

// first segment that starts at or after alignment column `col` (binary search over the device table, one
// 8-byte copy per probe: a reporting / verification helper, not on the transform path)
MsaPipeline::SegLoc MsaPipeline::locate(u64 col, hipStream_t st)
{
    if (!planned_) throw ParamError("locate needs a planned alignment");
    const u64 nseg = h_.nseg;
    auto at = [&](const u64* arr, u64 i) -> u64 {
        u64 v = 0;
        EDSX_HIP(hipMemcpyAsync(&v, arr + i, 8, hipMemcpyDeviceToHost, st));
        EDSX_HIP(hipStreamSynchronize(st));
        return v;
    };
    u64 lo = 0, hi = nseg;                                  // seg_start[nseg] == L
    while (lo < hi) {
        const u64 mid = lo + (hi - lo) / 2;
        if (at(seg_start_p_, mid) >= col) hi = mid; else lo = mid + 1;
    }
    SegLoc r{};
    r.seg = lo;
    r.col = lo < nseg ? at(seg_start_p_, lo) : h_.L;
    r.eds_off = lo < nseg ? at(eds_len_.as<u64>(), lo) : h_.E;
    r.seds_off = lo < nseg ? at(seds_len_.as<u64>(), lo) : h_.Q;
    return r;
}

void MsaPipeline::copy_columns(u64 col0, u64 ncols, uint8_t* host_out, hipStream_t st)
{
    if (!planned_) throw ParamError("copy_columns needs a planned alignment");
    if (col0 + ncols > h_.L || ncols == 0) throw ParamError("copy_columns: column range outside the alignment");
    const size_t bytes = (size_t)h_.S * ncols;
    colbuf_.ensure(bytes + 16);
    hipLaunchKernelGGL(k_copy_columns, dim3(256), dim3(256), 0, st, mv_, col0, ncols, colbuf_.as<uint8_t>());
    EDSX_HIP(hipMemcpyAsync(host_out, colbuf_.ptr, bytes, hipMemcpyDeviceToHost, st));
    EDSX_HIP(hipStreamSynchronize(st));
}

void MsaPipeline::emit(uint8_t* d_eds, uint8_t* d_seds, hipStream_t st)
{
    if (!planned_) throw ParamError("edsx_msa_emit_device called without a successful plan");
    EmitParams ep;
    ep.mv = mv_; ep.seg_start = seg_start_p_; ep.nseg_ptr = nseg_p_; ep.Hseg = hseg_p_; ep.segbase = segbase_p_;
    ep.eds_off = eds_len_.as<u64>(); ep.seds_off = seds_len_.as<u64>(); ep.eds = d_eds; ep.seds = d_seds;
    ep.nwords = h_.nwords;
    ep.list = nullptr; ep.list_n = nullptr;
    ep.stage_cols = stage_cols_; ep.stage_off = stage_off_;
    ep.scratch = seg_scratch_.as<uint8_t>(); ep.scratch_stride = seg_scratch_stride_;
    EDSX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_emit_variant<false>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)seg_lds_));
    auto launch_common = [&](hipStream_t s) {                 // the common segments: "{" reference text "}" and "{0}"
        TIMED("k_emit_common_seg", s, hipLaunchKernelGGL(k_emit_common_seg, dim3(4096), dim3(256), 0, s, mv_, seg_start_p_, nseg_p_,
                                                         eds_len_.as<u64>(), seds_len_.as<u64>(), d_eds, d_seds));
        TIMED("k_emit_common_long", s, hipLaunchKernelGGL(k_emit_common_long, dim3(512), dim3(256), 0, s, mv_, seg_start_p_,
                                                          eds_len_.as<u64>(), seds_len_.as<u64>(), long_list_.as<u64>(),
                                                          &hdr_.as<MsaHdr>()->long_n, d_eds, d_seds));
    };
    if (fast_) {
        FastParams fp = fp_;
        fp.eds = d_eds; fp.seds = d_seds;
        auto launch_emit = [&](auto kern, const char* name) {
            TIMED(name, st, hipLaunchKernelGGL(kern, dim3(persistent_grid(reinterpret_cast<const void*>(kern), 256, 0)),
                                               dim3(256), 0, st, fp));
        };
        // The main emitter fills the machine (beside it the small emitters only slow it down: measured in rounds 1 and 3);
        // behind it the wide emitter, the generic emitter (a few hundred workgroup-sized segments) and the common text
        // run side by side on three streams.
        launch_emit(k_emit_fast2, "k_emit_fast");
        ensure_side_streams();
        hipStream_t s1 = side_[0], s2 = side_[1];
#if defined(EDSX_EXPERIMENTS) && defined(EDSX_TAIL_SERIAL)
        s1 = st; s2 = st;                                   // every kernel alone: what each costs by itself
#endif
        EDSX_HIP(hipEventRecord(side_ev_[0], st));
        EDSX_HIP(hipStreamWaitEvent(s1, side_ev_[0], 0));
        EDSX_HIP(hipStreamWaitEvent(s2, side_ev_[0], 0));
        {
            auto launch_wide = [&](auto kern, const char* name) {
                TIMED(name, st, hipLaunchKernelGGL(kern, dim3(persistent_grid(reinterpret_cast<const void*>(kern), 256, 0)),
                                                   dim3(256), 0, st, fp));
            };
            if (h_.S >= 1000) {                                          // ids of five bytes exist
                launch_wide(k_emit_fast<true, true, 8>, "k_emit_fast_wide8"); launch_wide(k_emit_fast<true, true, 16>, "k_emit_fast_wide16");
            } else {
                launch_wide(k_emit_fast<false, true, 8>, "k_emit_fast_wide8"); launch_wide(k_emit_fast<false, true, 16>, "k_emit_fast_wide16");
            }
        }
        ep.gcache_stride = gc_stride_; ep.gcache_cap = gc_region_ / gc_stride_;
        ep.list = fp_.slow_list; ep.list_n = fp_.slow_count; ep.gcache = gcache_.as<uint8_t>();
        ep.list2 = fp_.slow_list2; ep.list2_n = fp_.slow_count2; ep.gcache2 = gcache_.as<uint8_t>() + gc_region_;
        TIMED("k_emit_variant_slow", s2, hipLaunchKernelGGL(k_emit_variant<false>, dim3(seg_grid()), dim3(GT), seg_lds_, s2, ep));
        launch_common(s1);
        EDSX_HIP(hipEventRecord(side_ev_[1], s1));
        EDSX_HIP(hipEventRecord(side_ev_[2], s2));
        EDSX_HIP(hipStreamWaitEvent(st, side_ev_[1], 0));
        EDSX_HIP(hipStreamWaitEvent(st, side_ev_[2], 0));
    } else {
        launch_common(st);
        RlParams rp = rl_;
        rp.eds = d_eds; rp.seds = d_seds;
        TIMED("k_rl_emit", st, hipLaunchKernelGGL(k_rl_emit, dim3(persistent_grid(reinterpret_cast<const void*>(k_rl_emit), 256, 0)),
                                                  dim3(256), 0, st, rp));
        ep.gcache_stride = gc_stride_; ep.gcache_cap = 2 * gc_region_ / gc_stride_; ep.gcache = gcache_.as<uint8_t>();
        ep.list = rl_.slow_list; ep.list_n = rl_.slow_count;
        if (big_) TIMED("k_emit_variant", st, hipLaunchKernelGGL(k_emit_variant<true>, dim3(seg_grid()), dim3(GT), seg_lds_, st, ep));
        else TIMED("k_emit_variant", st, hipLaunchKernelGGL(k_emit_variant<false>, dim3(seg_grid()), dim3(GT), seg_lds_, st, ep));
    }
    EDSX_HIP(hipGetLastError());
}

} // namespace edsx

