// multi_gpu.hip — see multi_gpu.hpp.  Host code (one thread per GPU) + one small kernel; the exchanges are RCCL calls.
#include "multi_gpu.hpp"

#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <thread>

namespace edsx {

// ---------------------------------------------------------------------------------------------------------------
// stitch plan (what edsparser_amd/multigpu.py::plan_stitch computes for the Python front end)
// ---------------------------------------------------------------------------------------------------------------
StitchPlan plan_stitch(const std::vector<SlabEdges>& e)
{
    const int n = (int)e.size();
    StitchPlan p;
    p.actions.resize(n);
    std::vector<char> join(n > 0 ? n - 1 : 0);
    for (int s = 0; s + 1 < n; s++) join[s] = e[s].last_is_variant == e[s + 1].first_is_variant;
    for (int s = 0; s + 1 < n;) {
        if (!join[s]) { s++; continue; }
        int b = s + 1;
        while (b < n - 1 && join[b] && e[b].n_segments == 1) b++;         // a slab that is ONE run passes the chain on
        p.chains.push_back(SlabChain{s, b, e[s].last_is_variant != 0});
        s = b;
    }
    for (int ci = 0; ci < (int)p.chains.size(); ci++) {
        const SlabChain& ch = p.chains[ci];
        if (!ch.variant) {              // common run: joined textually - "{" and "{0}" go on the right, "}" on the left
            for (int i = ch.first + 1; i <= ch.last; i++) { p.actions[i].front_eds += 1; p.actions[i].front_seds += 3; }
            for (int i = ch.first; i < ch.last; i++) p.actions[i].back_eds += 1;
        } else {                        // variant run: the left-most slab recomputes it from the raw columns
            p.actions[ch.first].back_eds += e[ch.first].last_eds_bytes;
            p.actions[ch.first].back_seds += e[ch.first].last_seds_bytes;
            p.actions[ch.first].owns.push_back(ci);
            for (int i = ch.first + 1; i <= ch.last; i++) {
                p.actions[i].front_eds += e[i].first_eds_bytes;
                p.actions[i].front_seds += e[i].first_seds_bytes;
            }
        }
    }
    return p;
}

// column range (col0, ncols) of slab `rank` inside a variant chain
static bool chain_columns(const SlabChain& ch, int rank, const std::vector<SlabEdges>& e, u64& col0, u64& ncols)
{
    if (rank < ch.first || rank > ch.last) return false;
    if (rank == ch.first) { col0 = e[rank].cols - e[rank].last_cols; ncols = e[rank].last_cols; }
    else { col0 = 0; ncols = e[rank].first_cols; }                        // (the whole slab when it is a single run)
    return true;
}

// ---------------------------------------------------------------------------------------------------------------
// FASTA geometry on the host (the row index of msa_transforms.cpp:46-68 for a plain uniform alignment)
// ---------------------------------------------------------------------------------------------------------------
MsaLayout msa_layout(const uint8_t* f, size_t n)
{
    MsaLayout lay;
    if (n == 0 || f[0] != '>') return lay;
    const uint8_t* he = static_cast<const uint8_t*>(memchr(f, '\n', n));
    if (!he) return lay;
    const u64 start0 = (u64)(he - f) + 1;
    if (start0 >= n) return lay;
    const uint8_t* nl = static_cast<const uint8_t*>(memchr(f + start0, '\n', n - start0));
    if (!nl) return lay;
    // first "\n>" behind the first header
    u64 h2 = n;
    for (const uint8_t* p = nl; p;) {
        const u64 i = (u64)(p - f);
        if (i + 1 < n && f[i + 1] == '>') { h2 = i; break; }
        if (i + 1 >= n) break;
        p = static_cast<const uint8_t*>(memchr(f + i + 1, '\n', n - i - 1));
    }
    if (h2 >= n) return lay;
    const u64 lw = (u64)(nl - f) - start0, draw = h2 - start0;
    if (lw == 0) return lay;
    u64 L, wrapped = 0;
    if (draw == lw) L = lw;
    else {
        const u64 nlines = (draw + 1 + lw) / (lw + 1);
        L = draw + 1 - nlines;
        wrapped = 1;
        if (L == 0 || (L - 1) / lw != nlines - 1) return lay;
    }
    u64 st = start0;
    while (true) {
        if (st + draw > n) return lay;
        lay.start.push_back(st);
        const u64 q = st + draw;
        if (q == n) break;
        if (f[q] != '\n') return lay;
        if (q + 1 == n) break;
        if (f[q + 1] != '>') {                           // blank lines at the very end only
            for (u64 i = q + 1; i < n; i++) if (f[i] != '\n') return lay;
            break;
        }
        const uint8_t* e = static_cast<const uint8_t*>(memchr(f + q + 1, '\n', n - q - 1));
        if (!e) return lay;
        st = (u64)(e - f) + 1;
    }
    if (lay.start.size() < 2) return lay;
    if (wrapped)                                          // every data line but a row's last has lw columns and then '\n'
        for (u64 s : lay.start)
            for (u64 o = lw; o < draw; o += lw + 1) if (f[s + o] != '\n') return lay;
    lay.draw = draw; lay.lw = wrapped ? lw : 0; lay.L = L; lay.ok = true;
    return lay;
}

// ---------------------------------------------------------------------------------------------------------------
// barrier of the rank threads, carrying the first failure to everyone
// ---------------------------------------------------------------------------------------------------------------
void RankBarrier::arrive(int rank, const std::string* failure)
{
    std::unique_lock<std::mutex> lk(mu_);
    if (failure && (!failed_ || rank < failed_rank_)) { failed_ = true; msg_ = *failure; failed_rank_ = rank; }
    const unsigned long g = gen_;
    if (++count_ == n_) { count_ = 0; gen_++; cv_.notify_all(); }
    else cv_.wait(lk, [&] { return gen_ != g; });
}

// ---------------------------------------------------------------------------------------------------------------
// exchanges
// ---------------------------------------------------------------------------------------------------------------
namespace {

class LocalExchange final : public Exchange {           // rank threads of one process, any device assignment
public:
    explicit LocalExchange(int n) : n_(n), bar_(n) {}
    void all_gather(int rank, const void* mine, size_t bytes, void* all) override
    {
        { std::lock_guard<std::mutex> g(mu_); if (stage_.size() < (size_t)n_ * bytes) stage_.resize((size_t)n_ * bytes); }
        bar_.arrive(rank, nullptr);                      // the staging area has its size
        std::memcpy(stage_.data() + (size_t)rank * bytes, mine, bytes);
        bar_.arrive(rank, nullptr);                      // every contribution is in
        std::memcpy(all, stage_.data(), (size_t)n_ * bytes);
        bar_.arrive(rank, nullptr);                      // everyone has read it: the area may be reused
    }
    const char* name() const override { return "in-process"; }
private:
    int n_; RankBarrier bar_; std::mutex mu_; std::vector<uint8_t> stage_;
};

#define EDSX_NCCL(call)                                                                       \
    do {                                                                                      \
        ncclResult_t r__ = (call);                                                            \
        if (r__ != ncclSuccess) throw DeviceError(std::string(#call) + ": " + ncclGetErrorString(r__)); \
    } while (0)

class RcclExchange final : public Exchange {            // one communicator, stream and pair of staging buffers per rank
public:
    explicit RcclExchange(const std::vector<int>& devices) : dev_(devices), comm_(devices.size()), st_(devices.size()),
                                                              send_(devices.size()), recv_(devices.size())
    {
        EDSX_NCCL(ncclCommInitAll(comm_.data(), (int)dev_.size(), dev_.data()));
        for (size_t r = 0; r < dev_.size(); r++) {
            EDSX_HIP(hipSetDevice(dev_[r]));
            EDSX_HIP(hipStreamCreateWithFlags(&st_[r], hipStreamNonBlocking));
        }
    }
    ~RcclExchange() override
    {
        for (size_t r = 0; r < dev_.size(); r++) {
            (void)hipSetDevice(dev_[r]);
            (void)hipStreamDestroy(st_[r]);
            (void)ncclCommDestroy(comm_[r]);
            send_[r].release(); recv_[r].release();
        }
    }
    void all_gather(int rank, const void* mine, size_t bytes, void* all) override
    {
        const size_t n = dev_.size();
        send_[rank].ensure(bytes + 16); recv_[rank].ensure(n * bytes + 16);
        EDSX_HIP(hipMemcpyAsync(send_[rank].ptr, mine, bytes, hipMemcpyHostToDevice, st_[rank]));
        EDSX_NCCL(ncclAllGather(send_[rank].ptr, recv_[rank].ptr, bytes, ncclUint8, comm_[rank], st_[rank]));
        EDSX_HIP(hipMemcpyAsync(all, recv_[rank].ptr, n * bytes, hipMemcpyDeviceToHost, st_[rank]));
        EDSX_HIP(hipStreamSynchronize(st_[rank]));
    }
    const char* name() const override { return "rccl"; }
private:
    std::vector<int> dev_; std::vector<ncclComm_t> comm_; std::vector<hipStream_t> st_;
    std::vector<DevBuf> send_, recv_;
};

// header of every row of a row image: ">r", blanks, newline; and the newline behind the row's columns
__global__ void k_row_frames(uint8_t* __restrict__ img, u64 nrows, u64 ncols, u64 h0, u64 h)
{
    const u64 wave = (blockIdx.x * (u64)blockDim.x + threadIdx.x) >> 6, nwaves = ((u64)gridDim.x * blockDim.x) >> 6;
    const u32 lane = threadIdx.x & 63;
    for (u64 r = wave; r < nrows; r += nwaves) {
        const u64 hl = r ? h : h0;
        uint8_t* p = img + (r ? h0 + (ncols + 1) + (r - 1) * (h + ncols + 1) : 0);
        for (u64 i = lane; i < hl; i += 64) p[i] = i == 0 ? '>' : i == 1 ? 'r' : i + 1 == hl ? '\n' : ' ';
        if (lane == 0) p[hl + ncols] = '\n';
    }
}

} // namespace

// ---------------------------------------------------------------------------------------------------------------
// Row image: columns [c0, c1) of every row of a host FASTA image as a one-line-per-row image in HBM whose rows all begin
// on multiples of 128 bytes (blank-padded headers).  The column scan reads every row in 128-byte pieces; from a
// 128-byte-aligned row those loads are aligned, and the scan of the 1000 x 100 Mb alignment takes 21.8 instead of
// 26.9 ms (round 3, bench.py "aligned_rows").  A row's slab is one contiguous piece of the file (one-line rows), so
// the upload is a 2D copy per stretch of equally spaced rows; wrapped rows are put together on the host.
// ---------------------------------------------------------------------------------------------------------------
RowImage upload_row_image(const uint8_t* fasta, const MsaLayout& lay, u64 c0, u64 c1, DevBuf& d_img, std::vector<uint8_t>& host_tmp,
                          hipStream_t st)
{
    RowImage ri;
    const u64 S = lay.start.size(), ncols = c1 - c0;
    ri.rows = S; ri.cols = ncols;
    ri.hdr0 = 128;
    ri.hdr = (128 - (ncols + 1) % 128) % 128;
    if (ri.hdr < 8) ri.hdr += 128;
    ri.bytes = ri.hdr0 + S * (ncols + 1) + (S - 1) * ri.hdr;
    d_img.ensure(ri.bytes + 16);
    uint8_t* img = d_img.as<uint8_t>();
    const u64 pitch = ri.hdr + ncols + 1;                      // rows 1 .. S-1
    auto data_off = [&](u64 r) { return r ? ri.hdr0 + (ncols + 1) + (r - 1) * pitch + ri.hdr : ri.hdr0; };
    if (lay.lw == 0) {
        // rows with headers of equal length are equally spaced in the file: one 2D copy per such stretch of rows
        EDSX_HIP(hipMemcpyAsync(img + data_off(0), fasta + lay.start[0] + c0, ncols, hipMemcpyHostToDevice, st));
        for (u64 s0 = 1; s0 < S;) {
            u64 s1 = s0;                                       // last row of the stretch
            const u64 sp = s0 + 1 < S ? lay.start[s0 + 1] - lay.start[s0] : pitch;
            while (s1 + 1 < S && lay.start[s1 + 1] - lay.start[s1] == sp) s1++;
            const u64 nrows = s1 - s0 + 1;
            EDSX_HIP(hipMemcpy2DAsync(img + data_off(s0), pitch, fasta + lay.start[s0] + c0, sp, ncols, nrows, hipMemcpyHostToDevice, st));
            s0 += nrows;
        }
    } else {
        // wrapped rows: columns [c0, c1) of a row are the bytes c + c / lw without the newlines between them
        host_tmp.resize(S * ncols);
        for (u64 s = 0; s < S; s++) {
            uint8_t* d = host_tmp.data() + s * ncols;
            const uint8_t* row = fasta + lay.start[s];
            for (u64 c = c0; c < c1;) {
                const u64 in_line = c % lay.lw, take = std::min<u64>(lay.lw - in_line, c1 - c);
                std::memcpy(d, row + c + c / lay.lw, take);
                d += take; c += take;
            }
        }
        EDSX_HIP(hipMemcpyAsync(img + data_off(0), host_tmp.data(), ncols, hipMemcpyHostToDevice, st));
        if (S > 1) EDSX_HIP(hipMemcpy2DAsync(img + data_off(1), pitch, host_tmp.data() + ncols, ncols, ncols, S - 1, hipMemcpyHostToDevice, st));
    }
    hipLaunchKernelGGL(k_row_frames, dim3(64), dim3(256), 0, st, img, S, ncols, ri.hdr0, ri.hdr);
    EDSX_HIP(hipGetLastError());
    return ri;
}

// ---------------------------------------------------------------------------------------------------------------
// MultiMsa
// ---------------------------------------------------------------------------------------------------------------
struct MultiMsa::Rank {
    int device = 0;
    MsaPipeline slab, mini;              // the slab's pipeline; boundary segments are recomputed through a second one
    DevBuf d_img, d_eds, d_seds, d_mini;
    std::vector<uint8_t> host_img;       // wrapped rows: the slab image is put together on the host
    std::string error;
};

MultiMsa::MultiMsa(const std::vector<int>& devices, bool use_rccl) : devices_(devices)
{
    if (devices.empty()) throw ParamError("edsx_multi_create: no devices");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) throw DeviceError("no usable gfx950 device (the engine has no CPU fallback)");
    for (int d : devices) if (d < 0 || d >= count) throw ParamError("edsx_multi_create: device id out of range");
    if (use_rccl) {
        std::vector<int> sorted = devices;
        std::sort(sorted.begin(), sorted.end());
        if (std::adjacent_find(sorted.begin(), sorted.end()) != sorted.end())
            throw ParamError("edsx_multi_create: RCCL needs one distinct device per rank (use the in-process exchange to share a device)");
        xch_.reset(new RcclExchange(devices));
    } else xch_.reset(new LocalExchange((int)devices.size()));
    bar_.reset(new RankBarrier((int)devices.size()));
    for (int d : devices) { ranks_.emplace_back(new Rank()); ranks_.back()->device = d; }
}

MultiMsa::~MultiMsa()
{
    for (auto& r : ranks_) { (void)hipSetDevice(r->device); r.reset(); }
}

void MultiMsa::transform(const uint8_t* fasta, size_t n, uint32_t context_len, HostBytes& eds, HostBytes& seds)
{
    const int N = world();
    if (n == 0) throw FormatError("Invalid MSA: empty input");
    MsaLayout lay;
    if (N > 1) lay = msa_layout(fasta, n);
    // (an l-EDS slab needs room for a standalone common run at either end: at least 4 l columns)
    partitioned_ = lay.ok && lay.L >= 2ull * (u64)N && lay.L / (u64)N >= 4ull * context_len;
    chains_ = 0;
    auto whole_on_rank0 = [&] {
        // One GPU: a file that is not a plain uniform alignment gets its error from the transform itself, in the reference's
        // words; an l-EDS whose slabs have no standalone common run to anchor the stitch on is not partitioned either.
        Rank& r0 = *ranks_[0];
        EDSX_HIP(hipSetDevice(r0.device));
        r0.d_img.ensure(n);
        EDSX_HIP(hipMemcpyAsync(r0.d_img.ptr, fasta, n, hipMemcpyHostToDevice, nullptr));
        uint64_t E = 0, Q = 0;
        r0.slab.plan(r0.d_img.as<uint8_t>(), n, context_len, nullptr, &E, &Q);
        r0.d_eds.ensure(E + 16); r0.d_seds.ensure(Q + 16);
        r0.slab.emit(r0.d_eds.as<uint8_t>(), r0.d_seds.as<uint8_t>(), nullptr);
        eds.take(E); seds.take(Q);
        PinnedDownload::copy(eds.data, r0.d_eds.ptr, E, nullptr);
        PinnedDownload::copy(seds.data, r0.d_seds.ptr, Q, nullptr);
    };
    if (!partitioned_) { whole_on_rank0(); return; }
    bar_->reset();
    no_anchor_ = false;
    piece_e_.assign(N, 0); piece_s_.assign(N, 0);
    std::vector<std::thread> th;
    auto one = [&](int r) { if (context_len) run_rank_leds(r, fasta, lay, context_len, eds, seds); else run_rank(r, fasta, n, lay, eds, seds); };
    for (int r = 1; r < N; r++) th.emplace_back([&, r] { one(r); });
    one(0);
    for (auto& t : th) t.join();
    if (bar_->failed()) {
        const std::string m = bar_->message();
        if (m.rfind("Invalid MSA", 0) == 0) throw FormatError(m);
        throw DeviceError(m);
    }
    if (no_anchor_) { partitioned_ = false; chains_ = 0; whole_on_rank0(); }
}

// One rank.  Every phase ends in the thread barrier, which also carries a failure of any rank to all of them: nobody
// enters a collective that a failed rank will not join.
void MultiMsa::run_rank(int r, const uint8_t* fasta, size_t n, const MsaLayout& lay, HostBytes& eds, HostBytes& seds)
{
    const int N = world();
    Rank& me = *ranks_[r];
    std::string fail;
    auto phase = [&](auto&& body) -> bool {               // false: some rank failed, leave
        if (fail.empty() && !bar_->failed()) {
            try { body(); } catch (const std::exception& ex) { fail = ex.what(); }
        }
        bar_->arrive(r, fail.empty() ? nullptr : &fail);
        return !bar_->failed();
    };
    const u64 S = lay.start.size(), L = lay.L;
    const u64 c0 = L * (u64)r / (u64)N, c1 = L * (u64)(r + 1) / (u64)N, ncols = c1 - c0;
    hipStream_t st = nullptr;
    uint64_t E = 0, Q = 0;
    SlabEdges mine{};
    std::vector<SlabEdges> edges(N);

    // ---- 1. slab image -> HBM, plan, emit, edge descriptors
    if (!phase([&] {
            EDSX_HIP(hipSetDevice(me.device));
            const RowImage ri = upload_row_image(fasta, lay, c0, c1, me.d_img, me.host_img, st);
            const uint8_t* img = me.d_img.as<uint8_t>();
            const u64 img_bytes = ri.bytes;
            me.slab.plan(img, img_bytes, 0, st, &E, &Q);
            me.d_eds.ensure(E + 16); me.d_seds.ensure(Q + 16);
            me.slab.emit(me.d_eds.as<uint8_t>(), me.d_seds.as<uint8_t>(), st);
            const MsaPipeline::Edges e = me.slab.edge_info(st);
            mine = SlabEdges{e.nseg, ncols, E, Q, e.fvar, e.fcols, e.feds, e.fseds, e.lvar, e.lcols, e.leds, e.lseds};
        })) return;

    // ---- 2. all-gather of the edge descriptors; the same plan on every rank
    if (!phase([&] { xch_->all_gather(r, &mine, sizeof(SlabEdges), edges.data()); })) return;
    const StitchPlan plan = plan_stitch(edges);
    const SlabAction& act = plan.actions[r];
    if (r == 0) chains_ = (int)plan.chains.size();
    std::vector<int> variant;
    for (int ci = 0; ci < (int)plan.chains.size(); ci++) if (plan.chains[ci].variant) variant.push_back(ci);

    // ---- 3. only when a variant run crosses a boundary: all-gather of the raw boundary columns (padded to the largest
    // contribution: every rank derives all block sizes from the edges), the owners recompute their segments
    std::vector<uint8_t> extra_e, extra_s;
    if (!variant.empty()) {
        struct Block { int chain; u64 col0, ncols; };
        auto blocks_of = [&](int rk) {
            std::vector<Block> b;
            for (int ci : variant) { u64 a, c; if (chain_columns(plan.chains[ci], rk, edges, a, c)) b.push_back(Block{ci, a, c}); }
            return b;
        };
        u64 cap = 1;
        for (int rk = 0; rk < N; rk++) { u64 sz = 0; for (const Block& b : blocks_of(rk)) sz += S * b.ncols; cap = std::max(cap, sz); }
        std::vector<uint8_t> buf(cap, 0), all((size_t)cap * N);
        if (!phase([&] {
                EDSX_HIP(hipSetDevice(me.device));
                u64 at = 0;
                for (const Block& b : blocks_of(r)) { me.slab.copy_columns(b.col0, b.ncols, buf.data() + at, st); at += S * b.ncols; }
            })) return;
        if (!phase([&] { xch_->all_gather(r, buf.data(), cap, all.data()); })) return;
        if (!phase([&] {
                EDSX_HIP(hipSetDevice(me.device));
                for (int ci : act.owns) {
                    const SlabChain& ch = plan.chains[ci];
                    u64 width = 0;
                    std::vector<std::pair<const uint8_t*, u64>> parts;        // row-major blocks of the same rows
                    for (int rk = ch.first; rk <= ch.last; rk++) {
                        u64 at = 0;
                        for (const Block& b : blocks_of(rk)) {
                            if (b.chain == ci) { parts.emplace_back(all.data() + (size_t)rk * cap + at, b.ncols); width += b.ncols; }
                            at += S * b.ncols;
                        }
                    }
                    std::vector<uint8_t> mini(S * (width + 4));
                    for (u64 s = 0; s < S; s++) {
                        uint8_t* d = mini.data() + s * (width + 4);
                        d[0] = '>'; d[1] = 'r'; d[2] = '\n'; d += 3;
                        for (auto& pr : parts) { std::memcpy(d, pr.first + s * pr.second, pr.second); d += pr.second; }
                        *d = '\n';
                    }
                    me.d_mini.ensure(mini.size() + 16);
                    EDSX_HIP(hipMemcpyAsync(me.d_mini.ptr, mini.data(), mini.size(), hipMemcpyHostToDevice, st));
                    uint64_t e2 = 0, q2 = 0;
                    me.mini.plan(me.d_mini.as<uint8_t>(), mini.size(), 0, st, &e2, &q2);
                    DevBuf oe, oq;
                    oe.ensure(e2 + 16); oq.ensure(q2 + 16);
                    me.mini.emit(oe.as<uint8_t>(), oq.as<uint8_t>(), st);
                    const size_t pe = extra_e.size(), pq = extra_s.size();
                    extra_e.resize(pe + e2); extra_s.resize(pq + q2);
                    EDSX_HIP(hipMemcpyAsync(extra_e.data() + pe, oe.ptr, e2, hipMemcpyDeviceToHost, st));
                    EDSX_HIP(hipMemcpyAsync(extra_s.data() + pq, oq.ptr, q2, hipMemcpyDeviceToHost, st));
                    EDSX_HIP(hipStreamSynchronize(st));
                }
            })) return;
    }

    // ---- 4. piece sizes (an all-gather only when some segment was recomputed), offsets, the output buffers
    auto bounds = [](const SlabEdges& e, const SlabAction& a, u64& elo, u64& ehi, u64& slo, u64& shi) {
        elo = a.front_eds; ehi = e.eds_bytes - std::min(e.eds_bytes, a.back_eds);
        slo = a.front_seds; shi = e.seds_bytes - std::min(e.seds_bytes, a.back_seds);
        if (ehi < elo) ehi = elo;
        if (shi < slo) shi = slo;
    };
    u64 elo, ehi, slo, shi;
    bounds(mine, act, elo, ehi, slo, shi);
    std::vector<u64> sizes(2 * (size_t)N);
    if (!variant.empty()) {
        const u64 my[2] = {(ehi - elo) + extra_e.size(), (shi - slo) + extra_s.size()};
        if (!phase([&] { xch_->all_gather(r, my, sizeof(my), sizes.data()); })) return;
    } else {
        for (int rk = 0; rk < N; rk++) {
            u64 a, b, c, d;
            bounds(edges[rk], plan.actions[rk], a, b, c, d);
            sizes[2 * rk] = b - a; sizes[2 * rk + 1] = d - c;
        }
    }
    u64 eoff = 0, soff = 0, etot = 0, stot = 0;
    for (int rk = 0; rk < N; rk++) {
        if (rk < r) { eoff += sizes[2 * rk]; soff += sizes[2 * rk + 1]; }
        etot += sizes[2 * rk]; stot += sizes[2 * rk + 1];
    }
    if (!phase([&] { if (r == 0) { eds.take(etot); seds.take(stot); } })) return;

    // ---- 5. every rank writes its piece at its offset
    phase([&] {
        EDSX_HIP(hipSetDevice(me.device));
        if (ehi > elo) PinnedDownload::copy(eds.data + eoff, me.d_eds.as<uint8_t>() + elo, ehi - elo, st);
        if (shi > slo) PinnedDownload::copy(seds.data + soff, me.d_seds.as<uint8_t>() + slo, shi - slo, st);
        if (!extra_e.empty()) std::memcpy(eds.data + eoff + (ehi - elo), extra_e.data(), extra_e.size());
        if (!extra_s.empty()) std::memcpy(seds.data + soff + (shi - slo), extra_s.data(), extra_s.size());
    });
}

// ---------------------------------------------------------------------------------------------------------------
// Context length l > 0 (parse_msa_to_leds_streaming).  An l-EDS joins variant runs with the common runs of fewer than l
// columns between them; only a common run of at least l columns (or one at either end of the alignment) stands alone
// (msa_transforms.cpp:133-190).  A slab transformed on its own applies the "at either end" clause at ITS ends, so its
// text is the alignment's text only between its first and its last standalone run of >= l columns - its anchors: what
// lies between two anchors depends on nothing outside them.  Every boundary is therefore recomputed from the last
// anchor of the left slab to the first anchor of the right one (both included: a mini alignment that begins and ends
// with a standalone run, transformed with the same l), by the left rank; the columns come through one all-gather.
//   slab r keeps   text[ behind its first anchor .. in front of its last anchor )      (rank 0 from its start, the
//   last rank to its end), followed by the text of the boundary r | r+1.
// A slab without two distinct anchors (or one whose anchors lie more than ANCHOR_MAX_COLS from its ends) makes every
// rank leave; the caller then transforms the whole image on one GPU.
// ---------------------------------------------------------------------------------------------------------------
namespace {
struct SlabAnchors {            // twelve u64, exchanged as they are
    u64 cols, eds_bytes, seds_bytes, ok;
    u64 tail_col, tail_eds, tail_seds;          // last anchor: its first column, text offsets in front of it
    u64 head_end, head_eds, head_seds;          // first anchor: the column behind it, text offsets behind it
    u64 pad0, pad1;
};
constexpr u64 ANCHOR_MAX_COLS = 1u << 16;
}

void MultiMsa::run_rank_leds(int r, const uint8_t* fasta, const MsaLayout& lay, uint32_t l, HostBytes& eds, HostBytes& seds)
{
    const int N = world();
    Rank& me = *ranks_[r];
    std::string fail;
    auto phase = [&](auto&& body) -> bool {
        if (fail.empty() && !bar_->failed()) {
            try { body(); } catch (const std::exception& ex) { fail = ex.what(); }
        }
        bar_->arrive(r, fail.empty() ? nullptr : &fail);
        return !bar_->failed();
    };
    const u64 S = lay.start.size(), L = lay.L;
    const u64 c0 = L * (u64)r / (u64)N, c1 = L * (u64)(r + 1) / (u64)N, ncols = c1 - c0;
    hipStream_t st = nullptr;
    uint64_t E = 0, Q = 0;
    SlabAnchors mine{};
    std::vector<SlabAnchors> all(N);

    // ---- 1. slab image -> HBM, plan + emit with the context length, anchors
    if (!phase([&] {
            EDSX_HIP(hipSetDevice(me.device));
            const RowImage ri = upload_row_image(fasta, lay, c0, c1, me.d_img, me.host_img, st);
            me.slab.plan(me.d_img.as<uint8_t>(), ri.bytes, l, st, &E, &Q);
            me.d_eds.ensure(E + 16); me.d_seds.ensure(Q + 16);
            me.slab.emit(me.d_eds.as<uint8_t>(), me.d_seds.as<uint8_t>(), st);
            const MsaPipeline::Anchors a = me.slab.anchor_info(l, st);
            mine.cols = ncols; mine.eds_bytes = E; mine.seds_bytes = Q;
            const bool need_head = r > 0, need_tail = r + 1 < N;
            bool ok = a.found != 0;
            if (ok && need_head && need_tail) ok = a.first_seg < a.last_seg;             // two distinct anchors
            if (ok && need_head) ok = a.first_end <= ANCHOR_MAX_COLS;
            if (ok && need_tail) ok = ncols - a.last_col <= ANCHOR_MAX_COLS;
            mine.ok = ok ? 1 : 0;
            if (ok) {
                mine.tail_col = need_tail ? a.last_col : ncols; mine.tail_eds = need_tail ? a.last_eds : E; mine.tail_seds = need_tail ? a.last_seds : Q;
                mine.head_end = need_head ? a.first_end : 0; mine.head_eds = need_head ? a.first_eds_end : 0; mine.head_seds = need_head ? a.first_seds_end : 0;
            }
        })) return;
    if (!phase([&] { xch_->all_gather(r, &mine, sizeof(SlabAnchors), all.data()); })) return;
    for (int rk = 0; rk < N; rk++) if (!all[rk].ok) { if (r == 0) no_anchor_ = true; return; }     // the same on every rank
    if (r == 0) chains_ = N - 1;

    // ---- 2. boundary columns: every rank contributes [tail_col, cols) and [0, head_end) of its slab (padded to the largest)
    auto tail_cols = [&](int rk) { return all[rk].cols - all[rk].tail_col; };
    u64 cap = 1;
    for (int rk = 0; rk < N; rk++) cap = std::max(cap, S * (tail_cols(rk) + all[rk].head_end));
    std::vector<uint8_t> buf(cap, 0), gathered((size_t)cap * N);
    if (!phase([&] {
            EDSX_HIP(hipSetDevice(me.device));
            if (tail_cols(r)) me.slab.copy_columns(mine.tail_col, tail_cols(r), buf.data(), st);
            if (mine.head_end) me.slab.copy_columns(0, mine.head_end, buf.data() + S * tail_cols(r), st);
        })) return;
    if (!phase([&] { xch_->all_gather(r, buf.data(), cap, gathered.data()); })) return;

    // ---- 3. the boundary r | r+1, recomputed by rank r
    std::vector<uint8_t> extra_e, extra_s;
    if (r + 1 < N) {
        if (!phase([&] {
                EDSX_HIP(hipSetDevice(me.device));
                const u64 wl = tail_cols(r), wr = all[r + 1].head_end, width = wl + wr;
                const uint8_t* left = gathered.data() + (size_t)r * cap;
                const uint8_t* right = gathered.data() + (size_t)(r + 1) * cap + S * tail_cols(r + 1);
                std::vector<uint8_t> mini(S * (width + 4));
                for (u64 s = 0; s < S; s++) {
                    uint8_t* d = mini.data() + s * (width + 4);
                    d[0] = '>'; d[1] = 'r'; d[2] = '\n'; d += 3;
                    std::memcpy(d, left + s * wl, wl); d += wl;
                    std::memcpy(d, right + s * wr, wr); d += wr;
                    *d = '\n';
                }
                me.d_mini.ensure(mini.size() + 16);
                EDSX_HIP(hipMemcpyAsync(me.d_mini.ptr, mini.data(), mini.size(), hipMemcpyHostToDevice, st));
                uint64_t e2 = 0, q2 = 0;
                me.mini.plan(me.d_mini.as<uint8_t>(), mini.size(), l, st, &e2, &q2);
                DevBuf oe, oq;
                oe.ensure(e2 + 16); oq.ensure(q2 + 16);
                me.mini.emit(oe.as<uint8_t>(), oq.as<uint8_t>(), st);
                extra_e.resize(e2); extra_s.resize(q2);
                EDSX_HIP(hipMemcpyAsync(extra_e.data(), oe.ptr, e2, hipMemcpyDeviceToHost, st));
                EDSX_HIP(hipMemcpyAsync(extra_s.data(), oq.ptr, q2, hipMemcpyDeviceToHost, st));
                EDSX_HIP(hipStreamSynchronize(st));
            })) return;
    } else if (!phase([] {})) return;

    // ---- 4. piece sizes, offsets, the output buffers; 5. every rank writes its piece at its offset
    const u64 elo = mine.head_eds, ehi = mine.tail_eds, slo = mine.head_seds, shi = mine.tail_seds;
    std::vector<u64> sizes(2 * (size_t)N);
    const u64 my[2] = {(ehi - elo) + extra_e.size(), (shi - slo) + extra_s.size()};
    if (!phase([&] { xch_->all_gather(r, my, sizeof(my), sizes.data()); })) return;
    u64 eoff = 0, soff = 0, etot = 0, stot = 0;
    for (int rk = 0; rk < N; rk++) {
        if (rk < r) { eoff += sizes[2 * rk]; soff += sizes[2 * rk + 1]; }
        etot += sizes[2 * rk]; stot += sizes[2 * rk + 1];
    }
    if (!phase([&] { if (r == 0) { eds.take(etot); seds.take(stot); } })) return;
    phase([&] {
        EDSX_HIP(hipSetDevice(me.device));
        if (ehi > elo) PinnedDownload::copy(eds.data + eoff, me.d_eds.as<uint8_t>() + elo, ehi - elo, st);
        if (shi > slo) PinnedDownload::copy(seds.data + soff, me.d_seds.as<uint8_t>() + slo, shi - slo, st);
        if (!extra_e.empty()) std::memcpy(eds.data + eoff + (ehi - elo), extra_e.data(), extra_e.size());
        if (!extra_s.empty()) std::memcpy(seds.data + soff + (shi - slo), extra_s.data(), extra_s.size());
    });
}

// ---------------------------------------------------------------------------------------------------------------
// Column batches on ONE GPU (edsx_msa_transform_batched, and what edsx_msa_transform falls back to when an alignment and
// its tables do not fit the device): the slabs of the multi-GPU path one after the other through one pipeline.  The
// working set is that of one slab (image, variant columns, records, tables); every slab's text goes to the host as soon
// as it is written, the boundaries are stitched at the end exactly as between ranks - but the boundary columns come
// straight from the host image, and the pieces are put together with host copies.
// ---------------------------------------------------------------------------------------------------------------
namespace {

// rows x columns [g0, g1) of the host image as a one-line-per-row alignment (">r\n" + the columns + "\n")
void host_mini_image(const uint8_t* fasta, const MsaLayout& lay, u64 g0, u64 g1, std::vector<uint8_t>& out)
{
    const u64 S = lay.start.size(), w = g1 - g0;
    out.resize(S * (w + 4));
    for (u64 s = 0; s < S; s++) {
        uint8_t* d = out.data() + s * (w + 4);
        d[0] = '>'; d[1] = 'r'; d[2] = '\n'; d += 3;
        const uint8_t* row = fasta + lay.start[s];
        if (lay.lw == 0) { std::memcpy(d, row + g0, w); d += w; }
        else
            for (u64 c = g0; c < g1;) {
                const u64 in_line = c % lay.lw, take = std::min<u64>(lay.lw - in_line, g1 - c);
                std::memcpy(d, row + c + c / lay.lw, take);
                d += take; c += take;
            }
        *d = '\n';
    }
}

} // namespace

bool msa_transform_batched(const BatchResources& R, const uint8_t* fasta, const MsaLayout& lay, uint32_t l, int K,
                           HostBytes& eds, HostBytes& seds, hipStream_t st)
{
    const u64 L = lay.L;
    if (!lay.ok || K < 2 || L < 2ull * (u64)K || L / (u64)K < 4ull * l) return false;
    std::vector<u64> c0(K + 1);
    for (int r = 0; r <= K; r++) c0[r] = L * (u64)r / (u64)K;
    std::vector<HostBytes> pe(K), ps(K);
    std::vector<SlabEdges> edges(K);
    std::vector<SlabAnchors> anch(K);

    // ---- 1. every slab: image -> HBM, plan, emit, descriptors, text -> host
    for (int r = 0; r < K; r++) {
        const u64 ncols = c0[r + 1] - c0[r];
        const RowImage ri = upload_row_image(fasta, lay, c0[r], c0[r + 1], *R.d_img, *R.host_tmp, st);
        uint64_t E = 0, Q = 0;
        R.slab->plan(R.d_img->as<uint8_t>(), ri.bytes, l, st, &E, &Q);
        R.d_eds->ensure(E + 16); R.d_seds->ensure(Q + 16);
        R.slab->emit(R.d_eds->as<uint8_t>(), R.d_seds->as<uint8_t>(), st);
        if (l == 0) {
            const MsaPipeline::Edges e = R.slab->edge_info(st);
            edges[r] = SlabEdges{e.nseg, ncols, E, Q, e.fvar, e.fcols, e.feds, e.fseds, e.lvar, e.lcols, e.leds, e.lseds};
        } else {
            const MsaPipeline::Anchors a = R.slab->anchor_info(l, st);
            SlabAnchors& m = anch[r];
            m = SlabAnchors{};
            m.cols = ncols; m.eds_bytes = E; m.seds_bytes = Q;
            const bool need_head = r > 0, need_tail = r + 1 < K;
            bool ok = a.found != 0;
            if (ok && need_head && need_tail) ok = a.first_seg < a.last_seg;
            if (ok && need_head) ok = a.first_end <= ANCHOR_MAX_COLS;
            if (ok && need_tail) ok = ncols - a.last_col <= ANCHOR_MAX_COLS;
            if (!ok) return false;                            // no anchors: not batched
            m.ok = 1;
            m.tail_col = need_tail ? a.last_col : ncols; m.tail_eds = need_tail ? a.last_eds : E; m.tail_seds = need_tail ? a.last_seds : Q;
            m.head_end = need_head ? a.first_end : 0; m.head_eds = need_head ? a.first_eds_end : 0; m.head_seds = need_head ? a.first_seds_end : 0;
        }
        pe[r].take(E); ps[r].take(Q);
        PinnedDownload::copy(pe[r].data, R.d_eds->ptr, E, st);
        PinnedDownload::copy(ps[r].data, R.d_seds->ptr, Q, st);
    }

    // ---- 2. boundaries: which bytes of every slab's text stay, and the text recomputed behind them
    std::vector<u64> elo(K), ehi(K), slo(K), shi(K);
    std::vector<std::vector<uint8_t>> xe(K), xs(K);
    std::vector<uint8_t> mini;
    auto recompute = [&](int owner, u64 g0, u64 g1) {
        host_mini_image(fasta, lay, g0, g1, mini);
        R.d_mini->ensure(mini.size() + 16);
        EDSX_HIP(hipMemcpyAsync(R.d_mini->ptr, mini.data(), mini.size(), hipMemcpyHostToDevice, st));
        uint64_t e2 = 0, q2 = 0;
        R.mini->plan(R.d_mini->as<uint8_t>(), mini.size(), l, st, &e2, &q2);
        DevBuf oe, oq;
        oe.ensure(e2 + 16); oq.ensure(q2 + 16);
        R.mini->emit(oe.as<uint8_t>(), oq.as<uint8_t>(), st);
        const size_t a = xe[owner].size(), b = xs[owner].size();
        xe[owner].resize(a + e2); xs[owner].resize(b + q2);
        EDSX_HIP(hipMemcpyAsync(xe[owner].data() + a, oe.ptr, e2, hipMemcpyDeviceToHost, st));
        EDSX_HIP(hipMemcpyAsync(xs[owner].data() + b, oq.ptr, q2, hipMemcpyDeviceToHost, st));
        EDSX_HIP(hipStreamSynchronize(st));
    };
    if (l == 0) {
        const StitchPlan plan = plan_stitch(edges);
        for (int r = 0; r < K; r++) {
            const SlabEdges& e = edges[r];
            const SlabAction& a = plan.actions[r];
            elo[r] = a.front_eds; ehi[r] = e.eds_bytes - std::min(e.eds_bytes, a.back_eds);
            slo[r] = a.front_seds; shi[r] = e.seds_bytes - std::min(e.seds_bytes, a.back_seds);
            if (ehi[r] < elo[r]) ehi[r] = elo[r];
            if (shi[r] < slo[r]) shi[r] = slo[r];
        }
        for (const SlabChain& ch : plan.chains)
            if (ch.variant)
                recompute(ch.first, c0[ch.first] + edges[ch.first].cols - edges[ch.first].last_cols, c0[ch.last] + edges[ch.last].first_cols);
    } else {
        for (int r = 0; r < K; r++) {
            elo[r] = anch[r].head_eds; ehi[r] = anch[r].tail_eds; slo[r] = anch[r].head_seds; shi[r] = anch[r].tail_seds;
            if (r + 1 < K) recompute(r, c0[r] + anch[r].tail_col, c0[r + 1] + anch[r + 1].head_end);
        }
    }

    // ---- 3. the pieces in order
    u64 etot = 0, stot = 0;
    for (int r = 0; r < K; r++) { etot += (ehi[r] - elo[r]) + xe[r].size(); stot += (shi[r] - slo[r]) + xs[r].size(); }
    eds.take(etot); seds.take(stot);
    u64 eo = 0, so = 0;
    for (int r = 0; r < K; r++) {
        std::memcpy(eds.data + eo, pe[r].data + elo[r], ehi[r] - elo[r]); eo += ehi[r] - elo[r];
        std::memcpy(seds.data + so, ps[r].data + slo[r], shi[r] - slo[r]); so += shi[r] - slo[r];
        if (!xe[r].empty()) { std::memcpy(eds.data + eo, xe[r].data(), xe[r].size()); eo += xe[r].size(); }
        if (!xs[r].empty()) { std::memcpy(seds.data + so, xs[r].data(), xs[r].size()); so += xs[r].size(); }
    }
    return true;
}

} // namespace edsx
