// capi.hip — the extern "C" boundary declared in include/edsx.h.  No exceptions cross it.
#include "../../include/edsx.h"

#include "genrandom.hpp"
#include "genvcf.hpp"
#include "merge_device.hpp"
#include "msa_device.hpp"
#include "multi_gpu.hpp"
#include "synth.hpp"
#include "vcf_device.hpp"

#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <vector>

using namespace edsx;

struct edsx_ctx {
    int device = 0;
    std::string err;
    MsaPipeline msa;
    MergePipeline merge;
    VcfPipeline vcf;
    GenPipeline gen;
    GenVcfPipeline genvcf;
    DevBuf d_in, d_eds, d_seds, synth_desc;
    std::vector<uint8_t> host_tmp;
    std::unique_ptr<MsaPipeline> mini;       // column batches: the boundary segments are recomputed through a second pipeline
    DevBuf d_mini;
    int last_batches = 0;                    // of the last edsx_msa_transform / _batched: 1 = one piece
};

namespace {

template <class F> int guarded(edsx_ctx* ctx, F&& f)
{
    if (!ctx) return EDSX_ERR_INVALID_PARAMETER;
    try {
        ctx->err.clear();
        hipError_t e = hipSetDevice(ctx->device);
        if (e != hipSuccess) throw DeviceError(std::string("hipSetDevice: ") + hipGetErrorString(e));
        f();
        return EDSX_OK;
    } catch (const FormatError& ex) { ctx->err = ex.what(); return EDSX_ERR_INVALID_FORMAT;
    } catch (const ParamError& ex) { ctx->err = ex.what(); return EDSX_ERR_INVALID_PARAMETER;
    } catch (const LimitError& ex) { ctx->err = ex.what(); return EDSX_ERR_BUILD_FAILED;
    } catch (const DeviceError& ex) { ctx->err = ex.what(); return EDSX_ERR_BUILD_FAILED;
    } catch (const std::bad_alloc&) { ctx->err = "out of host memory"; return EDSX_ERR_BUILD_FAILED;
    } catch (const std::exception& ex) { ctx->err = ex.what(); return EDSX_ERR_UNKNOWN; }
}

void take(edsx_buf* b, size_t n)
{
    b->data = HostBytes::alloc(n);
    b->size = n;
}

// everything the MSA entry points hold on the device goes back to the allocator
void release_msa_buffers(edsx_ctx* ctx)
{
    ctx->d_in.release(); ctx->d_eds.release(); ctx->d_seds.release(); ctx->d_mini.release();
    ctx->msa.~MsaPipeline();
    new (&ctx->msa) MsaPipeline();
    ctx->mini.reset();
}

// one piece: upload, plan, emit, download
void msa_transform_whole(edsx_ctx* ctx, const uint8_t* msa, size_t msa_size, const MsaLayout& lay, uint32_t context_len,
                         edsx_buf* eds, edsx_buf* seds)
{
    hipStream_t st = nullptr;
    // A plain uniform alignment of some size goes to HBM as a row image whose rows all begin on multiples of 128
    // bytes (2D copies: free on the way up): the column scan's loads are then aligned (multi_gpu.hip).  Everything
    // else - small inputs, files the geometry walk does not accept - is copied as it is, and the transform itself
    // words what is wrong with it.
    size_t dev_size = msa_size;
    if (lay.ok) dev_size = (size_t)upload_row_image(msa, lay, 0, lay.L, ctx->d_in, ctx->host_tmp, st).bytes;
    else {
        ctx->d_in.ensure(msa_size);
        EDSX_HIP(hipMemcpyAsync(ctx->d_in.ptr, msa, msa_size, hipMemcpyHostToDevice, st));
    }
    uint64_t E = 0, Q = 0;
    ctx->msa.plan(ctx->d_in.as<uint8_t>(), dev_size, context_len, st, &E, &Q);
    ctx->d_eds.ensure(E + 16);
    ctx->d_seds.ensure(Q + 16);
    ctx->msa.emit(ctx->d_eds.as<uint8_t>(), ctx->d_seds.as<uint8_t>(), st);
    take(eds, E);
    take(seds, Q);
    PinnedDownload::copy(eds->data, ctx->d_eds.ptr, E, st);
    PinnedDownload::copy(seds->data, ctx->d_seds.ptr, Q, st);
}

// K column batches; false: this input is not cut (see msa_transform_batched)
bool msa_transform_in_batches(edsx_ctx* ctx, const uint8_t* msa, const MsaLayout& lay, uint32_t context_len, int K,
                              edsx_buf* eds, edsx_buf* seds)
{
    if (!ctx->mini) ctx->mini.reset(new MsaPipeline());
    const BatchResources R{&ctx->msa, ctx->mini.get(), &ctx->d_in, &ctx->d_eds, &ctx->d_seds, &ctx->d_mini, &ctx->host_tmp};
    HostBytes e, q;
    if (!msa_transform_batched(R, msa, lay, context_len, K, e, q, nullptr)) return false;
    eds->size = e.size; eds->data = e.release();
    seds->size = q.size; seds->data = q.release();
    return true;
}

} // namespace

extern "C" {

const char* edsx_version(void) { return "edsx 0.1 (gfx950)"; }

int edsx_ctx_create(int device, edsx_ctx** out)
{
    if (!out) return EDSX_ERR_INVALID_PARAMETER;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count)
        return EDSX_ERR_BUILD_FAILED;           // no GPU: there is no CPU fallback
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return EDSX_ERR_BUILD_FAILED;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return EDSX_ERR_BUILD_FAILED;
    edsx_ctx* c = new (std::nothrow) edsx_ctx();
    if (!c) return EDSX_ERR_BUILD_FAILED;
    c->device = device;
    *out = c;
    return EDSX_OK;
}

void edsx_ctx_destroy(edsx_ctx* ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    delete ctx;
}

const char* edsx_last_error(const edsx_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

void edsx_buf_free(edsx_buf* buf)
{
    if (!buf) return;
    free(buf->data);
    buf->data = nullptr;
    buf->size = 0;
}

int edsx_msa_plan_device(edsx_ctx* ctx, const uint8_t* d_msa, size_t msa_size, uint32_t context_len,
                         void* stream, uint64_t* eds_bytes, uint64_t* seds_bytes)
{
    return guarded(ctx, [&] {
        if (!d_msa || !eds_bytes || !seds_bytes) throw ParamError("null argument");
        ctx->msa.plan(d_msa, msa_size, context_len, static_cast<hipStream_t>(stream), eds_bytes, seds_bytes);
        ctx->last_batches = 1;
    });
}

int edsx_msa_emit_device(edsx_ctx* ctx, uint8_t* d_eds, uint8_t* d_seds, void* stream)
{
    return guarded(ctx, [&] {
        if (!d_eds || !d_seds) throw ParamError("null argument");
        ctx->msa.emit(d_eds, d_seds, static_cast<hipStream_t>(stream));
    });
}

int edsx_msa_last_batches(const edsx_ctx* ctx) { return ctx ? ctx->last_batches : 0; }

int edsx_msa_last_info(const edsx_ctx* ctx, edsx_msa_info* info)
{
    // (after a transform in column batches the pipeline holds the plan of the LAST batch only: no info then)
    if (!ctx || !info || !ctx->msa.planned() || ctx->last_batches > 1) return EDSX_ERR_INVALID_PARAMETER;
    const MsaHdr& h = ctx->msa.header();
    info->n_rows = h.S; info->n_cols = h.L; info->line_width = h.lw ? h.lw : h.L;
    info->n_variant_cols = h.nv; info->n_segments = h.nseg; info->msa_bytes = ctx->msa.msa_bytes();
    info->n_slow_segments = h.slow_n + h.slow_n2;
    return EDSX_OK;
}

int edsx_msa_edge_info(edsx_ctx* ctx, edsx_msa_edges* out)
{
    return guarded(ctx, [&] {
        if (!out) throw ParamError("null argument");
        MsaPipeline::Edges e = ctx->msa.edge_info(nullptr);
        out->n_segments = e.nseg;
        out->first_is_variant = e.fvar; out->first_cols = e.fcols; out->first_eds_bytes = e.feds; out->first_seds_bytes = e.fseds;
        out->last_is_variant = e.lvar; out->last_cols = e.lcols; out->last_eds_bytes = e.leds; out->last_seds_bytes = e.lseds;
    });
}

int edsx_msa_anchor_info(edsx_ctx* ctx, uint64_t min_cols, edsx_msa_anchors* out)
{
    return guarded(ctx, [&] {
        if (!out) throw ParamError("null argument");
        const MsaPipeline::Anchors a = ctx->msa.anchor_info(min_cols, nullptr);
        out->n_segments = a.nseg; out->found = a.found; out->first_seg = a.first_seg; out->last_seg = a.last_seg;
        out->last_col = a.last_col; out->last_eds_bytes = a.last_eds; out->last_seds_bytes = a.last_seds;
        out->first_end = a.first_end; out->first_eds_end = a.first_eds_end; out->first_seds_end = a.first_seds_end;
    });
}

int edsx_msa_copy_columns(edsx_ctx* ctx, uint64_t col0, uint64_t ncols, uint8_t* host_out)
{
    return guarded(ctx, [&] {
        if (!host_out) throw ParamError("null argument");
        ctx->msa.copy_columns(col0, ncols, host_out, nullptr);
    });
}

int edsx_msa_locate_segment(edsx_ctx* ctx, uint64_t col, uint64_t* seg, uint64_t* seg_col, uint64_t* eds_off,
                            uint64_t* seds_off)
{
    return guarded(ctx, [&] {
        if (!seg || !seg_col || !eds_off || !seds_off) throw ParamError("null argument");
        const MsaPipeline::SegLoc r = ctx->msa.locate(col, nullptr);
        *seg = r.seg; *seg_col = r.col; *eds_off = r.eds_off; *seds_off = r.seds_off;
    });
}

struct edsx_multi_impl { std::unique_ptr<MultiMsa> m; std::string err; };

int edsx_multi_create(const int* device_ids, int n, int use_rccl, edsx_multi** out)
{
    if (!out) return EDSX_ERR_INVALID_PARAMETER;
    *out = nullptr;
    if (!device_ids || n <= 0) return EDSX_ERR_INVALID_PARAMETER;
    edsx_multi_impl* mi = new (std::nothrow) edsx_multi_impl();
    if (!mi) return EDSX_ERR_BUILD_FAILED;
    try {
        mi->m.reset(new MultiMsa(std::vector<int>(device_ids, device_ids + n), use_rccl != 0));
    } catch (const ParamError&) { delete mi; return EDSX_ERR_INVALID_PARAMETER;
    } catch (const std::exception&) { delete mi; return EDSX_ERR_BUILD_FAILED; }
    *out = reinterpret_cast<edsx_multi*>(mi);
    return EDSX_OK;
}
void edsx_multi_destroy(edsx_multi* m) { delete reinterpret_cast<edsx_multi_impl*>(m); }
const char* edsx_multi_last_error(const edsx_multi* m)
{
    return m ? reinterpret_cast<const edsx_multi_impl*>(m)->err.c_str() : "null handle";
}
int edsx_msa_transform_multi(edsx_multi* m, const uint8_t* msa, size_t msa_size, uint32_t context_len, edsx_buf* eds, edsx_buf* seds)
{
    if (eds) { eds->data = nullptr; eds->size = 0; }
    if (seds) { seds->data = nullptr; seds->size = 0; }
    edsx_multi_impl* mi = reinterpret_cast<edsx_multi_impl*>(m);
    if (!mi || !msa || !eds || !seds) return EDSX_ERR_INVALID_PARAMETER;
    try {
        mi->err.clear();
        HostBytes e, s;
        mi->m->transform(msa, msa_size, context_len, e, s);
        eds->size = e.size; eds->data = e.release();
        seds->size = s.size; seds->data = s.release();
        return EDSX_OK;
    } catch (const FormatError& ex) { mi->err = ex.what(); return EDSX_ERR_INVALID_FORMAT;
    } catch (const ParamError& ex) { mi->err = ex.what(); return EDSX_ERR_INVALID_PARAMETER;
    } catch (const LimitError& ex) { mi->err = ex.what(); return EDSX_ERR_BUILD_FAILED;
    } catch (const DeviceError& ex) { mi->err = ex.what(); return EDSX_ERR_BUILD_FAILED;
    } catch (const std::bad_alloc&) { mi->err = "out of host memory"; return EDSX_ERR_BUILD_FAILED;
    } catch (const std::exception& ex) { mi->err = ex.what(); return EDSX_ERR_UNKNOWN; }
}
int edsx_multi_last_partition(const edsx_multi* m, int* partitioned, int* chains)
{
    const edsx_multi_impl* mi = reinterpret_cast<const edsx_multi_impl*>(m);
    if (!mi) return EDSX_ERR_INVALID_PARAMETER;
    if (partitioned) *partitioned = mi->m->partitioned() ? 1 : 0;
    if (chains) *chains = mi->m->chains();
    return EDSX_OK;
}

void edsx_set_timing(edsx_ctx* ctx, int enabled) { if (ctx) ctx->msa.set_timing(enabled != 0); }
int edsx_get_timing(edsx_ctx* ctx, const char** names, float* total_ms, int* launches, int cap)
{
    return ctx ? ctx->msa.get_timing(names, total_ms, launches, cap) : 0;
}

int edsx_msa_transform(edsx_ctx* ctx, const uint8_t* msa, size_t msa_size, uint32_t context_len,
                       edsx_buf* eds, edsx_buf* seds)
{
    if (eds) { eds->data = nullptr; eds->size = 0; }
    if (seds) { seds->data = nullptr; seds->size = 0; }
    return guarded(ctx, [&] {
        if (!msa || !eds || !seds) throw ParamError("null argument");
        if (msa_size == 0) throw FormatError("Invalid MSA: empty input");
        MsaLayout lay;
        if (msa_size >= ((size_t)1 << 20)) lay = msa_layout(msa, msa_size);
        std::string oom;
        ctx->last_batches = 0;
        try { msa_transform_whole(ctx, msa, msa_size, lay, context_len, eds, seds); ctx->last_batches = 1; return; }
        catch (const OutOfDeviceMemory& ex) { oom = ex.what(); }
        // The alignment and its tables do not fit the device in one piece: column batches, each with the working set of
        // a K-th of the columns (the reference streams an alignment of any size, msa_transforms.cpp:36-90)
        edsx_buf_free(eds); edsx_buf_free(seds);
        for (int K = 2; K <= 256 && lay.ok; K *= 2) {
            release_msa_buffers(ctx);
            try {
                if (!msa_transform_in_batches(ctx, msa, lay, context_len, K, eds, seds)) break;
                ctx->last_batches = K;
                return;
            } catch (const OutOfDeviceMemory& ex) { oom = ex.what(); edsx_buf_free(eds); edsx_buf_free(seds); }
        }
        release_msa_buffers(ctx);
        throw LimitError("the alignment does not fit the device, in one piece or in column batches (" + oom + ")");
    });
}

int edsx_msa_transform_batched(edsx_ctx* ctx, const uint8_t* msa, size_t msa_size, uint32_t context_len, int batches,
                               edsx_buf* eds, edsx_buf* seds, int* batches_used)
{
    if (eds) { eds->data = nullptr; eds->size = 0; }
    if (seds) { seds->data = nullptr; seds->size = 0; }
    if (batches_used) *batches_used = 0;
    return guarded(ctx, [&] {
        if (!msa || !eds || !seds) throw ParamError("null argument");
        if (batches < 1) throw ParamError("edsx_msa_transform_batched: batches must be at least 1");
        if (msa_size == 0) throw FormatError("Invalid MSA: empty input");
        const MsaLayout lay = msa_layout(msa, msa_size);
        ctx->last_batches = 0;
        if (batches > 1 && lay.ok && msa_transform_in_batches(ctx, msa, lay, context_len, batches, eds, seds)) {
            if (batches_used) *batches_used = batches;
            ctx->last_batches = batches;
            return;
        }
        msa_transform_whole(ctx, msa, msa_size, lay, context_len, eds, seds);
        if (batches_used) *batches_used = 1;
        ctx->last_batches = 1;
    });
}

int edsx_leds_merge(edsx_ctx* ctx, const uint8_t* eds, size_t eds_size, const uint8_t* seds, size_t seds_size,
                    uint32_t context_len, int compact, edsx_buf* leds, edsx_buf* seds_out)
{
    if (leds) { leds->data = nullptr; leds->size = 0; }
    if (seds_out) { seds_out->data = nullptr; seds_out->size = 0; }
    return guarded(ctx, [&] {
        if (!leds || !seds_out || (!eds && eds_size)) throw ParamError("null argument");
        HostBytes out, sout;
        static const uint8_t none = 0;
        ctx->merge.run(eds ? eds : &none, eds_size, seds, seds_size, context_len, compact != 0, out, sout, nullptr);
        leds->size = out.size; leds->data = out.release();       // the download buffers themselves
        seds_out->size = sout.size; seds_out->data = sout.release();
    });
}

int edsx_eds_stats(edsx_ctx* ctx, const uint8_t* eds, size_t eds_size, const uint8_t* seds, size_t seds_size,
                   uint32_t context_len, edsx_eds_statistics* out)
{
    if (out) std::memset(out, 0, sizeof(*out));
    return guarded(ctx, [&] {
        if (!out || (!eds && eds_size)) throw ParamError("null argument");
        static const uint8_t none = 0;
        EdsStats s{};
        ctx->merge.stats(eds ? eds : &none, eds_size, seds, seds_size, context_len, s, nullptr);
        out->n_symbols = s.n_symbols; out->n_chars = s.n_chars; out->n_strings = s.n_strings;
        out->num_degenerate_symbols = s.num_degenerate; out->total_change_size = s.total_change_size;
        out->num_common_chars = s.num_common_chars; out->num_empty_strings = s.num_empty_strings;
        out->min_context_length = s.min_context; out->max_context_length = s.max_context;
        out->num_context_blocks = s.num_context_blocks;
        out->avg_context_length = s.num_context_blocks ? (double)s.num_common_chars / (double)s.num_context_blocks : 0.0;   // eds.cpp:428-432
        out->has_sources = s.has_sources; out->num_paths = s.num_paths; out->max_paths_per_string = s.max_paths_per_string;
        out->total_paths = s.total_paths;
        out->avg_paths_per_string = (s.has_sources && s.n_strings) ? (double)s.total_paths / (double)s.n_strings : 0.0;   // eds.cpp:501-503
        out->is_leds = (int)s.is_leds;
    });
}

int edsx_leds_tokenised_on_device(const edsx_ctx* ctx) { return ctx && ctx->merge.tokenised_on_device() ? 1 : 0; }

int edsx_leds_merge_range(edsx_ctx* ctx, const uint8_t* eds, size_t eds_size, const uint8_t* seds, size_t seds_size,
                          uint32_t context_len, int compact, int head_sentinel, int tail_sentinel, edsx_buf* leds,
                          edsx_buf* seds_out, int* head_intact, int* tail_intact)
{
    if (leds) { leds->data = nullptr; leds->size = 0; }
    if (seds_out) { seds_out->data = nullptr; seds_out->size = 0; }
    if (head_intact) *head_intact = 0;
    if (tail_intact) *tail_intact = 0;
    return guarded(ctx, [&] {
        if (!leds || !seds_out || !head_intact || !tail_intact || (!eds && eds_size)) throw ParamError("null argument");
        HostBytes out, sout;
        static const uint8_t none = 0;
        MergeShard sh;
        sh.head_sentinel = head_sentinel != 0; sh.tail_sentinel = tail_sentinel != 0;
        ctx->merge.run(eds ? eds : &none, eds_size, seds, seds_size, context_len, compact != 0, out, sout, nullptr, &sh);
        *head_intact = sh.head_intact ? 1 : 0;
        *tail_intact = sh.tail_intact ? 1 : 0;
        leds->size = out.size; leds->data = out.release();
        seds_out->size = sout.size; seds_out->data = sout.release();
    });
}

int edsx_vcf_tokenised_on_device(const edsx_ctx* ctx) { return ctx && ctx->vcf.tokenised_on_device() ? 1 : 0; }

int edsx_vcf_transform(edsx_ctx* ctx, const uint8_t* vcf, size_t vcf_size, const uint8_t* fasta, size_t fasta_size,
                       uint32_t context_len, edsx_buf* eds, edsx_buf* seds, edsx_vcf_stats* stats)
{
    if (eds) { eds->data = nullptr; eds->size = 0; }
    if (seds) { seds->data = nullptr; seds->size = 0; }
    if (stats) std::memset(stats, 0, sizeof(*stats));
    return guarded(ctx, [&] {
        if (!eds || !seds || (!vcf && vcf_size) || (!fasta && fasta_size)) throw ParamError("null argument");
        static const uint8_t none = 0;
        HostBytes e, s;
        VcfCounters c;
        try {
            ctx->vcf.run(vcf ? vcf : &none, vcf_size, fasta ? fasta : &none, fasta_size, e, s, c, nullptr);
        } catch (...) {
            if (stats) {   // the reference counts while parsing, before it can throw
                stats->total_variants = c.total_variants; stats->processed_variants = c.processed_variants;
                stats->skipped_malformed = c.skipped_malformed; stats->skipped_unsupported_sv = c.skipped_unsupported_sv;
            }
            throw;
        }
        if (stats) {
            stats->total_variants = c.total_variants; stats->processed_variants = c.processed_variants;
            stats->skipped_malformed = c.skipped_malformed; stats->skipped_unsupported_sv = c.skipped_unsupported_sv;
            stats->variant_groups = c.variant_groups;
        }
        if (context_len > 0) {   // vcf_transforms.cpp:735-755: EDS text -> LINEAR merge with defaults (compact)
            HostBytes lo, so;
            ctx->merge.run(e.data, e.size, s.data, s.size, context_len, true, lo, so, nullptr);
            eds->size = lo.size; eds->data = lo.release();
            seds->size = so.size; seds->data = so.release();
            return;
        }
        eds->size = e.size; eds->data = e.release();           // the download buffers themselves
        seds->size = s.size; seds->data = s.release();
    });
}

namespace {
void take_u64(edsx_buf* b, const std::vector<u64>& v)
{
    take(b, 8 * v.size());
    if (!v.empty()) std::memcpy(b->data, v.data(), 8 * v.size());
}
} // namespace

int edsx_vcf_index(edsx_ctx* ctx, const uint8_t* vcf, size_t vcf_size, edsx_buf* pos, edsx_buf* reflen, edsx_buf* line_off,
                   edsx_buf* line_len, edsx_vcf_stats* stats)
{
    for (edsx_buf* b : {pos, reflen, line_off, line_len}) if (b) { b->data = nullptr; b->size = 0; }
    if (stats) std::memset(stats, 0, sizeof(*stats));
    return guarded(ctx, [&] {
        if (!pos || !reflen || !line_off || !line_len || (!vcf && vcf_size)) throw ParamError("null argument");
        static const uint8_t none = 0;
        std::vector<u64> p, r, lo, ll;
        VcfCounters c;
        if (!ctx->vcf.index_device(vcf ? vcf : &none, vcf_size, nullptr, p, r, lo, ll, c))   // plain text: on the GPU
            vcf_index(vcf ? vcf : &none, vcf_size, p, r, lo, ll, c);
        if (stats) {
            stats->total_variants = c.total_variants; stats->processed_variants = c.processed_variants;
            stats->skipped_malformed = c.skipped_malformed; stats->skipped_unsupported_sv = c.skipped_unsupported_sv;
        }
        take_u64(pos, p); take_u64(reflen, r); take_u64(line_off, lo); take_u64(line_len, ll);
    });
}

int edsx_vcf_sort_order(const uint64_t* pos, size_t n, uint32_t* order_out)
{
    if ((!pos || !order_out) && n) return EDSX_ERR_INVALID_PARAMETER;
    if (n >= 0xffffffffull) return EDSX_ERR_INVALID_PARAMETER;
    static_assert(sizeof(u64) == sizeof(uint64_t), "u64");
    try { vcf_sort_order(reinterpret_cast<const u64*>(pos), n, order_out); } catch (...) { return EDSX_ERR_BUILD_FAILED; }
    return EDSX_OK;
}

int edsx_vcf_transform_range(edsx_ctx* ctx, const uint8_t* vcf, size_t vcf_size, const uint8_t* fasta, size_t fasta_size,
                             uint64_t cur0, uint64_t next_start, edsx_buf* eds, edsx_buf* seds, edsx_vcf_stats* stats)
{
    if (eds) { eds->data = nullptr; eds->size = 0; }
    if (seds) { seds->data = nullptr; seds->size = 0; }
    if (stats) std::memset(stats, 0, sizeof(*stats));
    return guarded(ctx, [&] {
        if (!eds || !seds || (!vcf && vcf_size) || (!fasta && fasta_size)) throw ParamError("null argument");
        static const uint8_t none = 0;
        HostBytes e, s;
        VcfCounters c;
        VcfRange range;
        range.presorted = true; range.cur0 = cur0; range.next_start = next_start;
        ctx->vcf.run(vcf ? vcf : &none, vcf_size, fasta ? fasta : &none, fasta_size, e, s, c, nullptr, range);
        if (stats) {
            stats->total_variants = c.total_variants; stats->processed_variants = c.processed_variants;
            stats->skipped_malformed = c.skipped_malformed; stats->skipped_unsupported_sv = c.skipped_unsupported_sv;
            stats->variant_groups = c.variant_groups;
        }
        eds->size = e.size; eds->data = e.release();
        seds->size = s.size; seds->data = s.release();
    });
}

int edsx_genrandomeds(edsx_ctx* ctx, uint64_t total_bp, double variability, uint32_t min_alt, uint32_t max_alt,
                      uint32_t var_len_max, double snp_ratio, const char* alphabet, uint64_t min_context, uint64_t seed,
                      edsx_buf* eds, edsx_buf* seds, uint64_t* n_sites)
{
    if (eds) { eds->data = nullptr; eds->size = 0; }
    if (seds) { seds->data = nullptr; seds->size = 0; }
    return guarded(ctx, [&] {
        if (!eds || !seds || !alphabet) throw ParamError("null argument");
        GenParams gp{};
        gp.total_bp = total_bp; gp.variability = variability; gp.min_alt = min_alt; gp.max_alt = max_alt;
        gp.var_len_max = var_len_max; gp.snp_ratio = snp_ratio; gp.min_context = min_context; gp.seed = seed;
        const size_t an = std::strlen(alphabet);
        if (an > 64) throw ParamError("Alphabet of more than 64 characters is not supported by this build");
        gp.alpha_n = (u32)an;
        std::memcpy(gp.alphabet, alphabet, an);
        HostBytes e, s;
        u64 sites = 0;
        ctx->gen.run(gp, e, s, sites, nullptr);
        if (n_sites) *n_sites = sites;
        eds->size = e.size; eds->data = e.release();
        seds->size = s.size; seds->data = s.release();
    });
}

int edsx_genvcf(edsx_ctx* ctx, uint64_t ref_len, uint64_t n_records, uint32_t n_samples, uint64_t seed, edsx_buf* vcf, edsx_buf* fasta)
{
    if (vcf) { vcf->data = nullptr; vcf->size = 0; }
    if (fasta) { fasta->data = nullptr; fasta->size = 0; }
    return guarded(ctx, [&] {
        if (!vcf || !fasta) throw ParamError("null argument");
        HostBytes v, f;
        ctx->genvcf.run(ref_len, n_records, n_samples, seed, v, f, nullptr);
        vcf->size = v.size; vcf->data = v.release();
        fasta->size = f.size; fasta->data = f.release();
    });
}

size_t edsx_msa_synth_size(uint32_t n_rows, uint64_t n_cols) { return synth_size(n_rows, n_cols); }

int edsx_msa_synth_device(edsx_ctx* ctx, uint8_t* d_out, size_t capacity, uint32_t n_rows,
                          uint64_t col0, uint64_t n_cols, double variant_fraction, uint64_t seed,
                          void* stream, size_t* written)
{
    return guarded(ctx, [&] {
        if (!d_out || n_rows < 1 || n_cols < 1) throw ParamError("bad synthetic alignment geometry");
        size_t need = synth_size(n_rows, n_cols);
        if (capacity < need) throw ParamError("output buffer too small for the synthetic alignment");
        ctx->synth_desc.ensure(8 * (size_t)n_cols);
        synth_generate(d_out, ctx->synth_desc.as<u64>(), n_rows, col0, n_cols, variant_fraction, seed,
                       static_cast<hipStream_t>(stream));
        EDSX_HIP(hipGetLastError());
        if (written) *written = need;
    });
}

size_t edsx_msa_synth_size_aligned(uint32_t n_rows, uint64_t n_cols, uint32_t row_align) { return synth_size_aligned(n_rows, n_cols, row_align); }

int edsx_msa_synth_device_aligned(edsx_ctx* ctx, uint8_t* d_out, size_t capacity, uint32_t n_rows, uint64_t col0, uint64_t n_cols,
                                  double variant_fraction, uint64_t seed, uint32_t row_align, void* stream, size_t* written)
{
    return guarded(ctx, [&] {
        if (!d_out || n_rows < 1 || n_cols < 1 || n_rows > 99999) throw ParamError("bad synthetic alignment geometry");
        if (row_align > 1 && (row_align & (row_align - 1))) throw ParamError("row alignment must be a power of two");
        size_t need = synth_size_aligned(n_rows, n_cols, row_align);
        if (capacity < need) throw ParamError("output buffer too small for the synthetic alignment");
        ctx->synth_desc.ensure(8 * (size_t)n_cols);
        synth_generate(d_out, ctx->synth_desc.as<u64>(), n_rows, col0, n_cols, variant_fraction, seed,
                       static_cast<hipStream_t>(stream), row_align);
        EDSX_HIP(hipGetLastError());
        if (written) *written = need;
    });
}

} // extern "C"
