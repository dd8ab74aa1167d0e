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
    if (context_len == 0 && N > 1) lay = msa_layout(fasta, n);
    partitioned_ = lay.ok && lay.L >= 2ull * (u64)N;
    chains_ = 0;
    if (!partitioned_) {
        // One GPU: a context length > 0 looks across runs (msa_transforms.cpp:133-190), and a file that is not a plain
        // uniform alignment gets its error from the transform itself, in the reference's words.
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
        return;
    }
    bar_->reset();
    piece_e_.assign(N, 0); piece_s_.assign(N, 0);
    std::vector<std::thread> th;
    for (int r = 1; r < N; r++) th.emplace_back([&, r] { run_rank(r, fasta, n, lay, eds, seds); });
    run_rank(0, fasta, n, lay, eds, seds);
    for (auto& t : th) t.join();
    if (bar_->failed()) {
        const std::string m = bar_->message();
        if (m.rfind("Invalid MSA", 0) == 0) throw FormatError(m);
        throw DeviceError(m);
    }
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

} // namespace edsx
