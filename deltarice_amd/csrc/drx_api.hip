// drx_api.hip -- C ABI (include/deltarice_hip.h) over the gfx950 kernels.
//
// Host-side glue only: option parsing, batch geometry, workspace ownership, launches.
// There is deliberately no CPU implementation of the codec here: every entry point
// either runs the HIP kernels or returns an error.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <chrono>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/deltarice_hip.h"
#include "drx_internal.h"

using namespace drx;

struct drx_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    SideStream side{nullptr, nullptr, nullptr};  // independent kernels of one call (launch_decode)
    int decode_impl = 8;   // launch_decode(): 8 = walk fused into the staged kernel, two samples per ring access
                           // (default), 7 = the same with a separate walk kernel, 5 / 1 = one sample per ring access
                           // (fused / separate walk), 0 = simple kernel
    int encode_impl = 2;  // 2: single pass, persistent (k_encode_stream) where the standard geometry applies; 1: single pass with
                          // look-back (k_encode_fused) everywhere; 0: size pass + scan + pack pass
    int profile = 0;      // bracket kernels with HIP events (drx_plan_last_timings)
    uint32_t debug_flags = 0;  // Geom::dbg
    std::string last_error;
    // scratch of the one-chunk host path (drx_filter_chunk_host), grown on demand
    void *d_raw = nullptr;   size_t raw_cap = 0;
    void *d_enc = nullptr;   size_t enc_cap = 0;
    uint64_t *d_off = nullptr;
    void *h_pin = nullptr;   size_t pin_cap = 0;
    void *h_stage = nullptr; size_t stage_cap = 0;  // drx_ctx_host_staging(): the direct-chunk HDF5 path's staging buffer
    // the host path's plan is kept between calls: HDF5 calls the filter once per chunk with the same
    // geometry, and creating a plan costs eight hipMalloc/hipFree pairs (about a millisecond)
    drx_plan *host_plan = nullptr;
    uint32_t hp_samples = 0, hp_L = 0, hp_k = 0, hp_ntaps = 0;
    int32_t hp_taps[DRX_MAX_TAPS] = {0};
    std::mutex mu;      // serialises drx_filter_chunk_host callers (HDF5 may call the filter from any thread)
    std::mutex err_mu;  // last_error
};

// Every entry point runs on the context's device and leaves the calling thread's current device as it found it
// (an application that works on another GPU must not find its device switched by an H5Dread).
struct DeviceGuard {
    int prev = -1;
    hipError_t err;
    explicit DeviceGuard(int device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        err = (prev == device) ? hipSuccess : hipSetDevice(device);
        if (prev == device) prev = -1;  // nothing to restore
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};
#define DRX_ON_DEVICE(ctx)                                                                         \
    DeviceGuard dev_guard_((ctx)->device);                                                         \
    if (dev_guard_.err != hipSuccess)                                                              \
        return fail((ctx), DRX_ERR_DEVICE, "hipSetDevice(%d) failed: %s", (ctx)->device, hipGetErrorString(dev_guard_.err))

struct drx_plan {
    drx_ctx *ctx = nullptr;
    Geom G{};
    uint64_t total_samples = 0;
    uint64_t max_words = 0;
    ChunkDesc *d_chunks = nullptr;
    uint32_t *d_wave_words = nullptr;  // n_i
    uint32_t *d_wave_rel = nullptr;    // encode: header position relative to chunk start
    uint64_t *d_wave_off = nullptr;    // decode: absolute header position
    uint64_t *d_chunk_words = nullptr;
    uint64_t *d_scan = nullptr;        // look-back state of the single-pass encoder + ticket
    uint32_t last_enc_path = 0;         // DRX_ENC_* of the last drx_encode
    uint64_t enc_words_per_wave = 0;   // of the plan's last encode (0: none yet)
    bool es_segs_ok = false;           // d_scan holds k_encode_stream_segs' state for this geometry (plan_alloc)
    int32_t *d_taps = nullptr;         // general prediction filter (nullptr: delta)
    uint32_t *d_seg_bits = nullptr;    // few long waveforms: bits and bit position of every 8192-sample segment,
    uint64_t *d_seg_pos = nullptr;     // allocated by the first encode that needs them
    uint64_t *d_seg_unit_base = nullptr;  // ragged plans the segment encoder takes
    uint32_t *d_pc_wg_base = nullptr;     // ragged plans the pieces encoder takes: first workgroup of every chunk
    uint64_t *d_pc_scan = nullptr;        // pieces encoder: look-back entry per workgroup + ticket
    uint64_t pc_wgs = 0;                  // its workgroups (0: the plan's geometry never takes it)
    void *d_pw = nullptr;              // a handful of chunks: candidate lists of the parallel header walk
    void *d_blk = nullptr;             // few waveforms: unit table, look-back state and flags of the block-parallel decoder
    uint32_t *d_walk_lists = nullptr;  // ragged plans: chunk indices, short-waveform chunks first
    uint2 *d_rag_order = nullptr;      // ragged plans: decode wavefronts, longest WaveformLength first
    uint32_t *d_iir_tab = nullptr;     // general filter behind the block decoder (drx_iir.hip): matrix tables of the plan's filter,
    uint64_t *d_iir_state = nullptr;   // look-back state of its tiles (+ ticket), and (ragged plans) the first tile of every chunk
    uint64_t *d_iir_chunk_base = nullptr;
    uint32_t *d_blk_list = nullptr;    // ragged plans the block decoder takes: waveform indices, class by class
    uint32_t n_short = 0, n_long = 0;
    DevStatus *d_status = nullptr;
    DevStatus *h_status = nullptr;  // pinned
    uint64_t *h_enc_words = nullptr;  // pinned, written by the encoders themselves (Geom::host_words): the last encode's word count, 0 = none yet
    bool last_was_encode = false;
    uint32_t last_path = 0;  // DRX_PATH_* of the last decode
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    bool ev_valid = false;
};

static drx_status fail(drx_ctx *ctx, drx_status st, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) {
        std::lock_guard<std::mutex> lock(ctx->err_mu);
        ctx->last_error = buf;
    }
    return st;
}

#define DRX_HIP(ctx, expr)                                                                      \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail((ctx), DRX_ERR_DEVICE, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

extern "C" {

const char *drx_version(void) { return "deltarice-hip 0.1 (gfx950)"; }

const char *drx_status_str(drx_status s) {
    switch (s) {
        case DRX_OK: return "ok";
        case DRX_ERR_ARG: return "invalid argument or compression_opts";
        case DRX_ERR_DEVICE: return "HIP device/runtime error";
        case DRX_ERR_CAPACITY: return "output capacity too small";
        case DRX_ERR_CORRUPT: return "corrupt encoded chunk";
        case DRX_ERR_UNSUPPORTED: return "not supported on the device path";
        case DRX_ERR_NOMEM: return "out of memory";
    }
    return "unknown";
}

int drx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// cd_values -> options; the meaning of each slot is the reference's (src/deltaRice.c:248-291).
drx_status drx_parse_cd_values(size_t cd_nelmts, const unsigned *cd_values, drx_opts *out) {
    if (!out || (cd_nelmts && !cd_values)) return DRX_ERR_ARG;
    long long m = 8;
    out->wave_len = -1;
    out->n_taps = 2;
    memset(out->taps, 0, sizeof out->taps);
    out->taps[0] = 1;
    out->taps[1] = -1;
    if (cd_nelmts >= 1) m = (int)cd_values[0];
    if (cd_nelmts >= 2) out->wave_len = (int)cd_values[1];
    if (cd_nelmts >= 3) {
        const long long nt = (int)cd_values[2];
        if (nt <= 0 || nt > DRX_MAX_TAPS || (size_t)nt + 3 > cd_nelmts) return DRX_ERR_ARG;
        out->n_taps = (uint32_t)nt;
        for (long long j = 0; j < nt; ++j) out->taps[j] = (int)cd_values[3 + j];
        if (out->taps[0] == 0) return DRX_ERR_ARG;
    }
    // M must be a power of two (src/deltaRice.c:114-136); 1 <= M <= 32768 is the range in
    // which the reference itself is well defined.
    if (m <= 0 || (m & (m - 1)) != 0 || m > 32768) return DRX_ERR_ARG;
    uint32_t k = 0;
    while ((1ll << k) != m) ++k;
    out->rice_k = k;
    if (out->wave_len == 0 || out->wave_len < -1) return DRX_ERR_ARG;
    return DRX_OK;
}

drx_status drx_ctx_create(int device, void *hip_stream, drx_ctx **out) {
    if (!out) return DRX_ERR_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return DRX_ERR_DEVICE;
    drx_ctx *c = new (std::nothrow) drx_ctx;
    if (!c) return DRX_ERR_NOMEM;
    c->device = device;
    DeviceGuard guard(device);
    if (guard.err != hipSuccess) { delete c; return DRX_ERR_DEVICE; }
    if (hip_stream) {
        c->stream = (hipStream_t)hip_stream;
    } else {
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return DRX_ERR_DEVICE; }
        c->own_stream = true;
    }
    // the side stream is an optimisation: a context without one runs everything on its stream
    if (hipStreamCreateWithFlags(&c->side.s, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->side.fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->side.join, hipEventDisableTiming) != hipSuccess) {
        if (c->side.s) (void)hipStreamDestroy(c->side.s);
        if (c->side.fork) (void)hipEventDestroy(c->side.fork);
        c->side = SideStream{nullptr, nullptr, nullptr};
        (void)hipGetLastError();
    }
    *out = c;
    return DRX_OK;
}

static void plan_free(drx_plan *p);

void drx_ctx_destroy(drx_ctx *c) {
    if (!c) return;
    DeviceGuard guard(c->device);
    if (c->host_plan) plan_free(c->host_plan);
    if (c->d_raw) (void)hipFree(c->d_raw);
    if (c->d_enc) (void)hipFree(c->d_enc);
    if (c->d_off) (void)hipFree(c->d_off);
    if (c->h_pin) (void)hipHostFree(c->h_pin);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    if (c->side.s) (void)hipStreamDestroy(c->side.s);
    if (c->side.fork) (void)hipEventDestroy(c->side.fork);
    if (c->side.join) (void)hipEventDestroy(c->side.join);
    delete c;
}

drx_status drx_ctx_synchronize(drx_ctx *c) {
    if (!c) return DRX_ERR_ARG;
    DRX_ON_DEVICE(c);
    DRX_HIP(c, hipStreamSynchronize(c->stream));
    return DRX_OK;
}

const char *drx_ctx_last_error(const drx_ctx *c) { return c ? c->last_error.c_str() : ""; }
void *drx_ctx_stream(const drx_ctx *c) { return c ? (void *)c->stream : nullptr; }
drx_status drx_ctx_host_staging(drx_ctx *ctx, size_t bytes, void **host_out) {
    if (!ctx || !host_out) return DRX_ERR_ARG;
    DRX_ON_DEVICE(ctx);
    if (ctx->stage_cap < bytes) {
        if (ctx->h_stage) { (void)hipStreamSynchronize(ctx->stream); (void)hipHostFree(ctx->h_stage); ctx->h_stage = nullptr; ctx->stage_cap = 0; }
        const size_t want = bytes + bytes / 8 + 4096;
        const hipError_t e = hipHostMalloc(&ctx->h_stage, want, hipHostMallocDefault);
        if (e != hipSuccess) return fail(ctx, DRX_ERR_NOMEM, "pinned staging buffer of %zu bytes: %s", want, hipGetErrorString(e));
        ctx->stage_cap = want;
    }
    *host_out = ctx->h_stage;
    return DRX_OK;
}
int drx_ctx_device(const drx_ctx *c) { return c ? c->device : -1; }

drx_status drx_ctx_set_option(drx_ctx *c, const char *key, int64_t value) {
    if (!c || !key) return DRX_ERR_ARG;
    if (!strcmp(key, "decode_impl")) {
#ifdef DRX_LEGACY
        if (value != 0 && value != 1 && value != 5 && value != 7 && value != 8) return DRX_ERR_ARG;
#else
        if (value != 0 && value != 7 && value != 8) return DRX_ERR_ARG;  // (1 / 5: one sample per ring access, -DDRX_LEGACY builds)
#endif
        c->decode_impl = (int)value;
        return DRX_OK;
    }
    if (!strcmp(key, "encode_impl")) {
        if (value < 0 || value > 2) return DRX_ERR_ARG;
        c->encode_impl = (int)value;
        return DRX_OK;
    }
    if (!strcmp(key, "debug_flags")) {
        c->debug_flags = (uint32_t)value;
        return DRX_OK;
    }
    if (!strcmp(key, "profile")) {
        c->profile = value != 0;
        return DRX_OK;
    }
    return DRX_ERR_ARG;
}

static void plan_free(drx_plan *p) {
    if (!p) return;
    DeviceGuard guard(p->ctx->device);
    if (p->d_chunks) (void)hipFree(p->d_chunks);
    if (p->d_wave_words) (void)hipFree(p->d_wave_words);
    if (p->d_wave_rel) (void)hipFree(p->d_wave_rel);
    if (p->d_wave_off) (void)hipFree(p->d_wave_off);
    if (p->d_chunk_words) (void)hipFree(p->d_chunk_words);
    if (p->d_scan) (void)hipFree(p->d_scan);
    if (p->d_taps) (void)hipFree(p->d_taps);
    if (p->d_walk_lists) (void)hipFree(p->d_walk_lists);
    if (p->d_rag_order) (void)hipFree(p->d_rag_order);
    if (p->d_iir_tab) (void)hipFree(p->d_iir_tab);
    if (p->d_iir_state) (void)hipFree(p->d_iir_state);
    if (p->d_iir_chunk_base) (void)hipFree(p->d_iir_chunk_base);
    if (p->d_blk_list) (void)hipFree(p->d_blk_list);
    if (p->d_seg_bits) (void)hipFree(p->d_seg_bits);
    if (p->d_seg_pos) (void)hipFree(p->d_seg_pos);
    if (p->d_seg_unit_base) (void)hipFree(p->d_seg_unit_base);
    if (p->d_pc_wg_base) (void)hipFree(p->d_pc_wg_base);
    if (p->d_pc_scan) (void)hipFree(p->d_pc_scan);
    if (p->d_pw) (void)hipFree(p->d_pw);
    if (p->d_blk) (void)hipFree(p->d_blk);
    if (p->d_status) (void)hipFree(p->d_status);
    if (p->h_status) (void)hipHostFree(p->h_status);
    if (p->h_enc_words) (void)hipHostFree(p->h_enc_words);
    for (hipEvent_t e : p->ev) if (e) (void)hipEventDestroy(e);
    delete p;
}

static drx_status plan_alloc(drx_ctx *ctx, drx_plan *p) {  // (callers hold the device guard)
    const uint64_t W = p->G.total_waves ? p->G.total_waves : 1;
    DRX_HIP(ctx, hipMalloc((void **)&p->d_wave_words, W * sizeof(uint32_t)));
    DRX_HIP(ctx, hipMalloc((void **)&p->d_wave_rel, W * sizeof(uint32_t)));
    DRX_HIP(ctx, hipMalloc((void **)&p->d_wave_off, W * sizeof(uint64_t)));
    DRX_HIP(ctx, hipMalloc((void **)&p->d_chunk_words, (p->G.n_chunks + 1) * sizeof(uint64_t)));
    // (k_encode_stream: size[W] | place[W] | control; its segment form: size[T] | place[2 T] | control with T tickets, at most
    // those of the shortest segments the dispatch may choose)
    uint64_t scan_words = 2 * W + 192;
    if (p->G.uniform && p->G.u_wave_len >= 64u) {
        const uint64_t tickets = W * es_seg_shape(p->G.u_wave_len, kEsSegMinLen).tpw;
        if (tickets < 0xffff0000ull) {
            p->es_segs_ok = true;
            if (3 * tickets + 192 > scan_words) scan_words = 3 * tickets + 192;
        }
    }
    DRX_HIP(ctx, hipMalloc((void **)&p->d_scan, scan_words * sizeof(uint64_t)));
    DRX_HIP(ctx, hipMemset(p->d_scan, 0, scan_words * sizeof(uint64_t)));
    DRX_HIP(ctx, hipMalloc((void **)&p->d_status, sizeof(DevStatus)));
    DRX_HIP(ctx, hipHostMalloc((void **)&p->h_status, sizeof(DevStatus), hipHostMallocDefault));
    memset(p->h_status, 0, sizeof(DevStatus));
    DRX_HIP(ctx, hipHostMalloc((void **)&p->h_enc_words, sizeof(uint64_t), hipHostMallocDefault));
    *p->h_enc_words = 0;
    p->G.host_words = p->h_enc_words;  // (pinned host memory is device-visible at the same address)
    for (hipEvent_t &e : p->ev) DRX_HIP(ctx, hipEventCreate(&e));
    return DRX_OK;
}

// Scratch that only some geometries need (segment encoder, long-waveform decoder, parallel header walks): allocated
// with the plan, so that drx_encode / drx_decode never allocate.
static drx_status plan_alloc_scratch(drx_ctx *ctx, drx_plan *p) {
    if (long_batch(p->G)) {
        const uint64_t units = long_batch_units(p->G);
        DRX_HIP(ctx, hipMalloc((void **)&p->d_seg_bits, units * sizeof(uint32_t)));
        DRX_HIP(ctx, hipMalloc((void **)&p->d_seg_pos, units * sizeof(uint64_t)));
    }
    if (const uint64_t nb = par_walk_scratch_bytes(p->G)) DRX_HIP(ctx, hipMalloc(&p->d_pw, nb));
    if (const uint64_t nb = blocks_scratch_bytes(p->G)) {
        DRX_HIP(ctx, hipMalloc(&p->d_blk, nb));
        // ... and, should the plan get a general prediction filter, the look-back state of the in-place inverse filter
        if (p->G.uniform) p->G.iir_n_tiles = iir_tiles(p->G, nullptr, nullptr);
        DRX_HIP(ctx, hipMalloc((void **)&p->d_iir_state, (p->G.iir_n_tiles + 1) * sizeof(uint64_t)));
        p->G.iir_state = p->d_iir_state;
    }
    if (p->G.uniform) {  // the pieces encoder's workgroups, where the geometry is one it can take (pieces_batch() decides per call)
        const uint32_t L = p->G.u_wave_len;
        const PieceShape sh = piece_shape(L, p->G.u_n_waves, p->G.k, piece_packable(L));
        const uint64_t wgs = (uint64_t)sh.wgs * p->G.n_chunks;
        if ((L >= kPcMinLen || piece_packable(L)) && (uint64_t)p->G.u_n_waves * sh.parts <= 0x7fffffffull && wgs <= 0x7fffffffull && p->total_samples >= 512u)
            p->pc_wgs = wgs;
    }
    if (p->pc_wgs) DRX_HIP(ctx, hipMalloc((void **)&p->d_pc_scan, pieces_scan_words(p->G, p->pc_wgs) * sizeof(uint64_t)));
    return DRX_OK;
}

drx_status drx_plan_create(drx_ctx *ctx, uint64_t n_chunks, const uint32_t *chunk_samples,
                           const uint32_t *chunk_wave_len, uint32_t rice_k, drx_plan **out) {
    if (!ctx || !out || !n_chunks || !chunk_samples || !chunk_wave_len) return DRX_ERR_ARG;
    *out = nullptr;
    if (rice_k > 15) return fail(ctx, DRX_ERR_ARG, "rice_k %u out of range 0..15", rice_k);
    if (n_chunks > 0xffffffffull) return fail(ctx, DRX_ERR_ARG, "too many chunks");
    DRX_ON_DEVICE(ctx);  // every allocation and table upload below, plan_alloc_scratch() included, lands on the context's device
    std::vector<ChunkDesc> desc(n_chunks);
    uint64_t soff = 0, wbase = 0, maxw = 0;
    bool uniform = true;
    for (uint64_t c = 0; c < n_chunks; ++c) {
        const uint32_t N = chunk_samples[c];
        // one chunk holds < 2^31 samples (int totalNumber, src/deltaRice.c:389)
        if (N == 0 || N > 0x7fffffffu) return fail(ctx, DRX_ERR_ARG, "chunk %llu: bad sample count %u", (unsigned long long)c, N);
        uint32_t L = chunk_wave_len[c] ? chunk_wave_len[c] : N;
        if (L > 0x7fffffffu) return fail(ctx, DRX_ERR_ARG, "chunk %llu: bad waveform length", (unsigned long long)c);
        const uint32_t W = (uint32_t)(((uint64_t)N + L - 1) / L);
        desc[c] = ChunkDesc{soff, wbase, N, L, W, 0};
        if (N != chunk_samples[0] || L != (chunk_wave_len[0] ? chunk_wave_len[0] : chunk_samples[0])) uniform = false;
        soff += N;
        wbase += W;
        // 25 bits per sample worst case, rounded up per waveform, + headers
        maxw += 1 + W + (((uint64_t)N * 25u + 31u) >> 5) + W;
    }
    drx_plan *p = new (std::nothrow) drx_plan;
    if (!p) return DRX_ERR_NOMEM;
    p->ctx = ctx;
    p->total_samples = soff;
    p->max_words = maxw;
    p->G.n_chunks = n_chunks;
    p->G.total_waves = wbase;
    p->G.uniform = uniform ? 1u : 0u;
    p->G.u_n_samples = desc[0].n_samples;
    p->G.u_wave_len = desc[0].wave_len;
    p->G.u_n_waves = desc[0].n_waves;
    p->G.k = rice_k;
    p->G.total_samples = soff;
    drx_status st = plan_alloc(ctx, p);
    if (st == DRX_OK && !uniform) {
        hipError_t e = hipMalloc((void **)&p->d_chunks, n_chunks * sizeof(ChunkDesc));
        if (e == hipSuccess) e = hipMemcpy(p->d_chunks, desc.data(), n_chunks * sizeof(ChunkDesc), hipMemcpyHostToDevice);
        // walk lists: chunks of short waveforms are walked through LDS, the others hop by hop
        std::vector<uint32_t> lists;
        for (uint64_t c = 0; c < n_chunks; ++c) if (desc[c].wave_len <= kWalkShortLenHost) lists.push_back((uint32_t)c);
        p->n_short = (uint32_t)lists.size();
        for (uint64_t c = 0; c < n_chunks; ++c) if (desc[c].wave_len > kWalkShortLenHost) lists.push_back((uint32_t)c);
        p->n_long = (uint32_t)lists.size() - p->n_short;
        if (e == hipSuccess) e = hipMalloc((void **)&p->d_walk_lists, lists.size() * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMemcpy(p->d_walk_lists, lists.data(), lists.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
        uint32_t max_groups = 0;
        for (uint64_t c = 0; c < n_chunks; ++c) max_groups = std::max(max_groups, (desc[c].n_waves + 63u) / 64u);
        p->G.walk_short = p->d_walk_lists;
        p->G.walk_long = p->d_walk_lists ? p->d_walk_lists + p->n_short : nullptr;
        p->G.n_short = p->n_short;
        p->G.n_long = p->n_long;
        p->G.max_groups = max_groups;
        {
            uint32_t max_len = 0;
            for (uint64_t c = 0; c < n_chunks; ++c) max_len = std::max(max_len, desc[c].wave_len);
            p->G.max_wave_len64 = 64ull * max_len;
        }
        // decode order: wavefronts (groups of 64 waveforms of one chunk) by decreasing WaveformLength
        {
            std::vector<uint32_t> by_len(n_chunks);
            for (uint64_t c = 0; c < n_chunks; ++c) by_len[c] = (uint32_t)c;
            std::stable_sort(by_len.begin(), by_len.end(), [&](uint32_t a, uint32_t b) { return desc[a].wave_len > desc[b].wave_len; });
            std::vector<uint2> order;
            uint64_t n_long_groups = 0;
            for (uint32_t c : by_len)
                for (uint32_t j = 0; j < (desc[c].n_waves + 63u) / 64u; ++j) {
                    order.push_back(make_uint2(c, j));
                    if (desc[c].wave_len > kWalkShortLenHost) ++n_long_groups;
                }
            if (order.size() <= 0x7fffffffull) {
                if (e == hipSuccess) e = hipMalloc((void **)&p->d_rag_order, order.size() * sizeof(uint2));
                if (e == hipSuccess) e = hipMemcpy(p->d_rag_order, order.data(), order.size() * sizeof(uint2), hipMemcpyHostToDevice);
                p->G.rag_order = p->d_rag_order;
                p->G.rag_groups = (uint32_t)order.size();
                p->G.rag_groups_long = (uint32_t)n_long_groups;
            }
        }
        // parallel header walks for small ragged batches: every long-waveform chunk within the chunk-wide walk's
        // capacity, every short-waveform chunk worth the two block passes (same limits as for uniform batches)
        {
            bool ok = p->n_long <= kPwMaxChunks && p->n_short <= kPwMaxChunks;
            uint64_t bmax = 0;
            uint32_t min_len = 0xffffffffu, min_long_waves = 0xffffffffu;
            for (uint64_t c = 0; c < n_chunks && ok; ++c) {
                const ChunkDesc &d = desc[c];
                if (d.wave_len > kWalkShortLenHost) {
                    ok = d.n_waves <= kPwMaxWaves;
                    min_long_waves = std::min(min_long_waves, d.n_waves);
                } else {
                    ok = d.wave_len >= 16u && p->n_short <= d.n_waves / 35u;
                    const uint64_t mw = 1u + 2ull * d.n_waves + (((uint64_t)d.n_samples * 25u + 31u) >> 5);
                    bmax = std::max<uint64_t>(bmax, (mw + 4095u) / 4096u);
                    min_len = std::min(min_len, d.wave_len);
                }
            }
            p->G.rag_par = ok && bmax <= 0xfffffu;
            p->G.rag_bw_blocks_max = (uint32_t)bmax;
            p->G.rag_bw_min_len = min_len;
            p->G.rag_pw_min_waves = min_long_waves;
        }
        // the segment encoder for ragged batches with short (<= 2048) or long (>= 16384) waveforms somewhere: unit
        // (waveform x 8192-sample segment slot) numbering per chunk
        bool seg = false;
        std::vector<uint64_t> ub(n_chunks + 1, 0);
        for (uint64_t c = 0; c < n_chunks; ++c) {
            seg = seg || desc[c].wave_len <= kSegShortLenHost || desc[c].wave_len >= kSegLongLenHost;
            ub[c + 1] = ub[c] + (uint64_t)desc[c].n_waves * ((desc[c].wave_len + 8191u) / 8192u);
        }
        if (seg) {
            if (e == hipSuccess) e = hipMalloc((void **)&p->d_seg_unit_base, ub.size() * sizeof(uint64_t));
            if (e == hipSuccess) e = hipMemcpy(p->d_seg_unit_base, ub.data(), ub.size() * sizeof(uint64_t), hipMemcpyHostToDevice);
            p->G.seg_unit_base = p->d_seg_unit_base;
            p->G.seg_units = ub[n_chunks];
        }
        // the pieces encoder (drx_pieces.hip): every WaveformLength within its range and some chunk of short or of long
        // waveforms -- or every WaveformLength above its range (waveforms over several workgroups); workgroup numbering per chunk
        {
            bool ok = soff >= 512u, some = false, all_super = true, none_super = true, all_packed = true;
            for (uint64_t c = 0; c < n_chunks; ++c) all_packed = all_packed && piece_packable(desc[c].wave_len);
            std::vector<uint32_t> wb(n_chunks + 1, 0);
            uint64_t wgs = 0;
            for (uint64_t c = 0; c < n_chunks && ok; ++c) {
                const PieceShape sh = piece_shape(desc[c].wave_len, desc[c].n_waves, p->G.k, all_packed);
                ok = (all_packed || desc[c].wave_len >= kPcMinLen) && (uint64_t)desc[c].n_waves * sh.parts <= 0x7fffffffull;
                all_super = all_super && sh.parts > 1u;
                none_super = none_super && sh.parts == 1u;
                some = some || all_packed || sh.run > 1u || sh.segs > 1u;
                wgs += sh.wgs;
                ok = ok && wgs <= 0x7fffffffull;
                wb[c + 1] = (uint32_t)wgs;
            }
            if (ok && some && (all_super || none_super)) {
                if (e == hipSuccess) e = hipMalloc((void **)&p->d_pc_wg_base, wb.size() * sizeof(uint32_t));
                if (e == hipSuccess) e = hipMemcpy(p->d_pc_wg_base, wb.data(), wb.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
                p->G.pc_wg_base = p->d_pc_wg_base;
                p->G.pc_super = all_super ? 1u : 0u;
                p->G.pc_packed = all_packed ? 1u : 0u;
                p->pc_wgs = wgs;
            }
        }
        // few long waveforms: the block-parallel decoder (and, behind it, the in-place inverse of a general filter)
        std::vector<uint32_t> blk_list(wbase ? wbase : 1);
        blocks_plan_ragged(p->G, desc.data(), blk_list.data());
        if (p->G.rag_blocks) {
            if (e == hipSuccess) e = hipMalloc((void **)&p->d_blk_list, blk_list.size() * sizeof(uint32_t));
            if (e == hipSuccess) e = hipMemcpy(p->d_blk_list, blk_list.data(), blk_list.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
            p->G.rag_blk_list = p->d_blk_list;
            std::vector<uint64_t> tb(n_chunks + 1, 0);
            p->G.iir_n_tiles = iir_tiles(p->G, desc.data(), tb.data());
            if (e == hipSuccess) e = hipMalloc((void **)&p->d_iir_chunk_base, tb.size() * sizeof(uint64_t));
            if (e == hipSuccess) e = hipMemcpy(p->d_iir_chunk_base, tb.data(), tb.size() * sizeof(uint64_t), hipMemcpyHostToDevice);
            p->G.iir_chunk_tile_base = p->d_iir_chunk_base;
        }
        if (e != hipSuccess) st = fail(ctx, DRX_ERR_DEVICE, "chunk table upload failed: %s", hipGetErrorString(e));
    }
    if (st != DRX_OK) { plan_free(p); return st; }
    p->G.chunks = p->d_chunks;
    if ((st = plan_alloc_scratch(ctx, p)) != DRX_OK) { plan_free(p); return st; }
    *out = p;
    return DRX_OK;
}

drx_status drx_plan_create_uniform(drx_ctx *ctx, uint64_t n_chunks, uint32_t chunk_samples,
                                   uint32_t wave_len, uint32_t rice_k, drx_plan **out) {
    if (!ctx || !out || !n_chunks) return DRX_ERR_ARG;
    *out = nullptr;
    if (rice_k > 15) return fail(ctx, DRX_ERR_ARG, "rice_k %u out of range 0..15", rice_k);
    if (chunk_samples == 0 || chunk_samples > 0x7fffffffu) return fail(ctx, DRX_ERR_ARG, "bad chunk sample count");
    if (n_chunks > 0xffffffffull) return fail(ctx, DRX_ERR_ARG, "too many chunks");
    const uint32_t L = wave_len ? wave_len : chunk_samples;
    if (L > 0x7fffffffu) return fail(ctx, DRX_ERR_ARG, "bad waveform length");
    DRX_ON_DEVICE(ctx);  // (as in drx_plan_create)
    const uint32_t W = (uint32_t)(((uint64_t)chunk_samples + L - 1) / L);
    drx_plan *p = new (std::nothrow) drx_plan;
    if (!p) return DRX_ERR_NOMEM;
    p->ctx = ctx;
    p->total_samples = n_chunks * chunk_samples;
    p->max_words = n_chunks * (1 + 2ull * W + (((uint64_t)chunk_samples * 25u + 31u) >> 5));
    p->G.chunks = nullptr;
    p->G.n_chunks = n_chunks;
    p->G.total_waves = n_chunks * W;
    p->G.uniform = 1;
    p->G.u_n_samples = chunk_samples;
    p->G.u_wave_len = L;
    p->G.u_n_waves = W;
    p->G.k = rice_k;
    p->G.total_samples = p->total_samples;
    drx_status st = plan_alloc(ctx, p);
    if (st == DRX_OK) st = plan_alloc_scratch(ctx, p);
    if (st != DRX_OK) { plan_free(p); return st; }
    *out = p;
    return DRX_OK;
}

drx_status drx_plan_set_filter(drx_plan *p, uint32_t n_taps, const int32_t *taps) {
    if (!p || !taps || n_taps == 0 || n_taps > DRX_MAX_TAPS) return DRX_ERR_ARG;
    drx_ctx *ctx = p->ctx;
    if (taps[0] == 0) return fail(ctx, DRX_ERR_ARG, "taps[0] must not be 0 (the inverse filter divides by it)");
    DRX_ON_DEVICE(ctx);
    DRX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    p->enc_words_per_wave = 0;  // (another filter, another code length: what the last encode measured no longer applies)
    if (p->h_enc_words) *p->h_enc_words = 0;
    p->G.fast_taps = 0;
    p->G.enc_fast = 0;
    if (n_taps == 2 && taps[0] == 1 && taps[1] == -1) {  // checkIfDeltaFilter, src/deltaRice.c:38-46
        p->G.n_taps = 0;
        p->G.taps = nullptr;
        p->G.iir_tab = nullptr;
        p->G.blk_iir_tab = nullptr;
        return DRX_OK;
    }
    if (n_taps <= 4) {
        p->G.enc_fast = 1;
        for (uint32_t j = 0; j < 4; ++j) p->G.enc_t[j] = (j < n_taps) ? (uint32_t)taps[j] & 0xffffu : 0u;
    }
    if (n_taps <= 4 && (taps[0] == 1 || taps[0] == -1)) {
        p->G.fast_taps = 1;
        p->G.fast_t0neg = taps[0] == -1;
        for (uint32_t j = 1; j < 4; ++j) p->G.fast_nt[j - 1] = (j < n_taps) ? 0u - (uint32_t)taps[j] : 0u;
    }
    p->G.iir_tab = nullptr;
    p->G.blk_iir_tab = nullptr;
    if (p->G.fast_taps && p->d_iir_state) {  // the block decoder's geometry: the inverse filter's matrix tables
        const uint32_t words = kIirTabWords + blocks_iir_tab_words();  // k_iir_tiles' tables, then the block decoder's own
        std::vector<uint32_t> tab(words);
        iir_tables(p->G.fast_nt, p->G.fast_t0neg, tab.data());
        blocks_iir_tables(p->G.fast_nt, p->G.fast_t0neg, tab.data() + kIirTabWords);
        if (!p->d_iir_tab) DRX_HIP(ctx, hipMalloc((void **)&p->d_iir_tab, words * sizeof(uint32_t)));
        DRX_HIP(ctx, hipMemcpy(p->d_iir_tab, tab.data(), words * sizeof(uint32_t), hipMemcpyHostToDevice));
        p->G.iir_tab = p->d_iir_tab;
        p->G.blk_iir_tab = p->d_iir_tab + kIirTabWords;
    }
    if (!p->d_taps) DRX_HIP(ctx, hipMalloc((void **)&p->d_taps, DRX_MAX_TAPS * sizeof(int32_t)));
    DRX_HIP(ctx, hipMemcpy(p->d_taps, taps, n_taps * sizeof(int32_t), hipMemcpyHostToDevice));
    p->G.n_taps = n_taps;
    p->G.taps = p->d_taps;
    return DRX_OK;
}

void drx_plan_destroy(drx_plan *p) { plan_free(p); }
uint64_t drx_plan_n_chunks(const drx_plan *p) { return p ? p->G.n_chunks : 0; }
uint64_t drx_plan_total_samples(const drx_plan *p) { return p ? p->total_samples : 0; }
uint64_t drx_plan_total_waves(const drx_plan *p) { return p ? p->G.total_waves : 0; }
uint64_t drx_plan_max_encoded_words(const drx_plan *p) { return p ? p->max_words : 0; }
const uint32_t *drx_plan_wave_words(const drx_plan *p) { return p ? p->d_wave_words : nullptr; }
const uint64_t *drx_plan_wave_word_off(const drx_plan *p) { return p ? p->d_wave_off : nullptr; }
uint32_t drx_plan_last_decode_path(const drx_plan *p) { return p ? p->last_path : 0u; }
uint32_t drx_plan_last_encode_path(const drx_plan *p) { return p ? p->last_enc_path : 0u; }

drx_status drx_plan_read_wave_words(drx_plan *p, uint32_t *host_out) {
    if (!p || !host_out) return DRX_ERR_ARG;
    drx_ctx *ctx = p->ctx;
    DRX_ON_DEVICE(ctx);
    DRX_HIP(ctx, hipMemcpyAsync(host_out, p->d_wave_words, p->G.total_waves * sizeof(uint32_t),
                                hipMemcpyDeviceToHost, ctx->stream));
    DRX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return DRX_OK;
}

// k_encode_stream or k_encode_fused?  The persistent form wins where a wavefront's ring (kEsRingWords) holds a waveform's
// code AND most of the next one's (the next is coded while the first waits for its place), and where the batch feeds its
// 4096 wavefronts a few waveforms each -- measured (profiles/r04_notes.md section 1e), stream / fused: 100 chunks of 14 M
// samples at WaveformLength 3600 ... 8192 (1060-1660 words per waveform) 1.02-1.70 / 1.09-1.97 ms, at 10000 (2020 words: the
// next waveform cannot start) 1.72 / 1.25; AR(1) under m = 4 (1740 words, 11 % escapes) 6.44 / 5.70; one chunk of 2000
// waveforms 0.031 / 0.023, ten chunks 0.126 / 0.137.  A waveform's words are not known before it is coded: the plan's last
// encode says (the same data shape comes again), and before that k + 3.5 bits per sample (the RiceParameter that suits).
static bool stream_encoder_suits(const drx_plan *p) {
    const Geom &G = p->G;
    if (fused_wide(G) != 0 || G.total_waves < 8192u) return false;
    const uint64_t L = G.uniform ? G.u_wave_len : G.max_wave_len64 / 64u;
    const uint64_t words = p->enc_words_per_wave ? p->enc_words_per_wave : (L * (2u * G.k + 7u)) / 64u;
    return words <= (uint64_t)kEsRingWords * 67u / 100u;
}

// k_encode_stream_segs or k_encode_pieces?  Long waveforms in a batch that feeds the persistent grid's 4096 wavefronts a few
// segments each; returns the segment length to aim at (0: not this encoder) -- what leaves a ring room for most of the next
// segment (57 % of it, as a 7000-sample waveform of the headline does), from the bits per sample of the plan's last encode or,
// before that, k + 3.5.  Debug flag 4194304: wherever the geometry allows, in segments of kEsSegMinLen samples (tests).
constexpr uint64_t kEsSegsFromLen = 20480;  // WaveformLengths from here on take the segment form (16 384: k_encode_pieces 0.61 ms, this 0.66; 24 000: 0.69 / 0.61)
static uint32_t stream_segs_target(const drx_plan *p, uint32_t dbg, int encode_impl) {
    const Geom &G = p->G;
    if (encode_impl != 2 || !p->es_segs_ok || !(G.n_taps == 0 || G.enc_fast) || (dbg & 4096u)) return 0u;
    if (dbg & 4194304u) return kEsSegMinLen;
    const uint64_t L = G.u_wave_len;
    uint64_t min_len = kEsSegsFromLen;
#ifdef DRX_ABLATION
    if (const char *e = getenv("DRX_SEGS_MIN_LEN")) min_len = (uint64_t)atoll(e);  // (A/B builds: where the two encoders cross)
#endif
    if (L < min_len) return 0u;
    const uint64_t bps16 = p->enc_words_per_wave ? (p->enc_words_per_wave * 512u) / L : 16u * G.k + 56u;  // bits per sample x 16
    uint64_t t = ((uint64_t)kEsRingWords * 32u * 57u / 100u) * 16u / (bps16 ? bps16 : 1u);
    t = t > kEsSegMaxLen ? kEsSegMaxLen : (t < kEsSegMinLen ? kEsSegMinLen : t);
    const EsSegShape sh = es_seg_shape((uint32_t)L, (uint32_t)t);
    if (G.total_waves * sh.nseg < 8192u) return 0u;
    return (uint32_t)t;
}

drx_status drx_encode(drx_plan *p, const int16_t *d_in, uint32_t *d_out, uint64_t out_cap_words,
                      uint64_t *d_chunk_word_off) {
    if (!p || !d_in || !d_out || !d_chunk_word_off) return DRX_ERR_ARG;
    drx_ctx *ctx = p->ctx;
    DRX_ON_DEVICE(ctx);
    DRX_HIP(ctx, hipMemsetAsync(p->d_status, 0, sizeof(DevStatus), ctx->stream));
    p->G.dbg = ctx->debug_flags;
    // what the plan's last encode measured decides this one's kernel -- read from the word the encoders write to pinned host
    // memory, so that callers that never wait for an encode (bench.py's steps) are covered without a copy or an event
    if (const uint64_t w = *(volatile uint64_t *)p->h_enc_words) p->enc_words_per_wave = p->G.total_waves ? w / p->G.total_waves : 0;
    const bool single = ctx->encode_impl >= 1;
    if (const uint32_t seg_target = stream_segs_target(p, ctx->debug_flags, ctx->encode_impl)) {
        DRX_HIP(ctx, launch_encode_stream_segs(p->G, seg_target, d_in, d_out, out_cap_words, d_chunk_word_off, p->d_wave_words,
                                               p->d_scan, p->d_status, ctx->profile ? p->ev : nullptr, ctx->stream));
        p->last_enc_path = DRX_ENC_STREAM_SEGS;
    } else if (single && p->d_pc_scan && pieces_batch(p->G)) {
        DRX_HIP(ctx, launch_encode_pieces(p->G, d_in, p->total_samples, d_out, out_cap_words, d_chunk_word_off, p->d_wave_words,
                                          p->d_pc_scan, p->pc_wgs, p->d_status, ctx->profile ? p->ev : nullptr, ctx->stream));
        p->last_enc_path = DRX_ENC_PIECES;
    } else if (single && long_batch(p->G) && !(ctx->debug_flags & 256u)) {
        if (!p->d_seg_bits) {  // only when a diagnostic debug_flags value forces this path on a geometry that does not take it
            const uint64_t units = long_batch_units(p->G);
            DRX_HIP(ctx, hipMalloc((void **)&p->d_seg_bits, units * sizeof(uint32_t)));
            DRX_HIP(ctx, hipMalloc((void **)&p->d_seg_pos, units * sizeof(uint64_t)));
        }
        DRX_HIP(ctx, launch_encode_long(p->G, d_in, d_out, out_cap_words, d_chunk_word_off, p->d_wave_words, p->d_wave_rel,
                                        p->d_chunk_words, p->d_seg_bits, p->d_seg_pos, p->d_status,
                                        ctx->profile ? p->ev : nullptr, ctx->stream));
        p->last_enc_path = DRX_ENC_SEGMENTS;
    } else if (ctx->encode_impl == 2 && (p->G.n_taps == 0 || p->G.enc_fast) && (stream_encoder_suits(p) || (ctx->debug_flags & 524288u))) {
        DRX_HIP(ctx, launch_encode_stream(p->G, d_in, d_out, out_cap_words, d_chunk_word_off, p->d_wave_words,
                                          p->d_scan, p->d_status, ctx->profile ? p->ev : nullptr, ctx->stream));
        p->last_enc_path = DRX_ENC_STREAM;
    } else if (single && (p->G.n_taps == 0 || p->G.enc_fast)) {
        DRX_HIP(ctx, launch_encode_fused(p->G, d_in, d_out, out_cap_words, d_chunk_word_off, p->d_wave_words,
                                         p->d_scan, p->d_status, ctx->profile ? p->ev : nullptr, ctx->stream));
        p->last_enc_path = DRX_ENC_FUSED;
    } else {
        DRX_HIP(ctx, launch_encode(p->G, d_in, d_out, out_cap_words, d_chunk_word_off, p->d_wave_words,
                                   p->d_wave_rel, p->d_chunk_words, p->d_status, ctx->profile ? p->ev : nullptr,
                                   ctx->stream));
        p->last_enc_path = DRX_ENC_TWO_PASS;
    }
    p->ev_valid = ctx->profile != 0;
    p->last_was_encode = true;
    return DRX_OK;
}

static drx_status decode_launch(drx_plan *p, const uint32_t *d_in, uint64_t in_words,
                                const uint64_t *d_chunk_word_off, int16_t *d_out, bool tables_ready,
                                const uint32_t *d_sideband = nullptr) {
    if (!p || !d_in || !d_chunk_word_off || !d_out) return DRX_ERR_ARG;
    drx_ctx *ctx = p->ctx;
    DRX_ON_DEVICE(ctx);
    DRX_HIP(ctx, hipMemsetAsync(p->d_status, 0, sizeof(DevStatus), ctx->stream));
    p->G.dbg = ctx->debug_flags;
    // (in_words says nothing about how long a waveform's code is -- it may be, and in bench.py IS, the buffer's capacity; round 4
    // took it for the stream's length for a while and sent the headline batch to k_encode_fused)
    if (d_sideband) {  // header positions from the caller's n_i table, checked against the stream (k_sideband_tables)
        DRX_HIP(ctx, launch_sideband_tables(p->G, d_in, in_words, d_chunk_word_off, d_sideband, p->d_wave_off, p->d_wave_words,
                                            p->d_status, ctx->stream));
        tables_ready = true;
    }
    // (d_pw / d_blk were allocated with the plan; a plan whose filter was set to general taps afterwards simply does
    // not take those paths)
    DRX_HIP(ctx, launch_decode(p->G, d_in, in_words, d_chunk_word_off, d_out, p->d_wave_off,
                               p->d_wave_words, p->d_scan, p->d_status,
                               (tables_ready ? 100 : 0) + ctx->decode_impl, p->d_pw, p->d_blk, ctx->side.s ? &ctx->side : nullptr,
                               ctx->profile ? p->ev : nullptr, ctx->stream, &p->last_path));
    p->ev_valid = ctx->profile != 0;
    p->last_was_encode = false;
    return DRX_OK;
}

drx_status drx_decode(drx_plan *p, const uint32_t *d_in, uint64_t in_words,
                      const uint64_t *d_chunk_word_off, int16_t *d_out) {
    return decode_launch(p, d_in, in_words, d_chunk_word_off, d_out, false);
}

drx_status drx_decode_with_wave_words(drx_plan *p, const uint32_t *d_in, uint64_t in_words, const uint64_t *d_chunk_word_off,
                                      const uint32_t *d_wave_words, int16_t *d_out) {
    if (!d_wave_words) return DRX_ERR_ARG;
    if (p && d_wave_words == p->d_wave_words) return fail(p->ctx, DRX_ERR_ARG, "the side-band table must not be the plan's own (copy it first)");
    return decode_launch(p, d_in, in_words, d_chunk_word_off, d_out, false, d_wave_words);
}

// Header chain of ONE encoded chunk in host memory (src/deltaRice.c:320-325), with the validation the device
// walk does: sample count, every n_i between 1 + k and 25 bits per sample, the chain ends exactly at the chunk end.
static bool walk_chunk_host(const uint32_t *w, uint64_t n_words, uint32_t n_samples, uint32_t wave_len, uint32_t k,
                            uint64_t *wave_off, uint32_t *wave_words) {
    if (n_words < 2 || w[0] != n_samples) return false;
    const uint32_t L = wave_len ? wave_len : n_samples;
    const uint64_t W = ((uint64_t)n_samples + L - 1) / L;
    uint64_t at = 1;
    for (uint64_t i = 0; i < W; ++i) {
        if (at >= n_words) return false;
        const uint32_t n = w[at];
        const uint64_t len = (i + 1 == W) ? (uint64_t)n_samples - i * L : L;
        if (n > ((len * 25u + 31u) >> 5) || n < ((len * (k + 1u) + 31u) >> 5) || at + 1u + n > n_words) return false;
        wave_off[i] = at;
        wave_words[i] = n;
        at += (uint64_t)n + 1u;
    }
    return at == n_words;
}

drx_status drx_estimate_words(drx_plan *p, const int16_t *d_in, uint64_t words_out[16]) {
    if (!p || !d_in || !words_out) return DRX_ERR_ARG;
    drx_ctx *ctx = p->ctx;
    DRX_ON_DEVICE(ctx);
    // the look-back scratch is idle outside drx_encode/drx_decode: its first 16 words hold the sums
    DRX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    DRX_HIP(ctx, launch_estimate_words(p->G, d_in, (unsigned long long *)p->d_scan, ctx->stream));
    DRX_HIP(ctx, hipMemcpyAsync(words_out, p->d_scan, 16 * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    DRX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return DRX_OK;
}

drx_status drx_plan_last_timings(drx_plan *p, float ms[4]) {
    if (!p || !ms) return DRX_ERR_ARG;
    drx_ctx *ctx = p->ctx;
    if (!p->ev_valid) return fail(ctx, DRX_ERR_ARG, "set the context option \"profile\" before the call to be timed");
    DRX_ON_DEVICE(ctx);
    DRX_HIP(ctx, hipEventSynchronize(p->ev[3]));
    for (int i = 0; i < 3; ++i) DRX_HIP(ctx, hipEventElapsedTime(&ms[i], p->ev[i], p->ev[i + 1]));
    DRX_HIP(ctx, hipEventElapsedTime(&ms[3], p->ev[0], p->ev[3]));
    return DRX_OK;
}

drx_status drx_plan_finish(drx_plan *p, uint64_t *total_words) {
    if (!p) return DRX_ERR_ARG;
    drx_ctx *ctx = p->ctx;
    DRX_ON_DEVICE(ctx);
    DRX_HIP(ctx, hipMemcpyAsync(p->h_status, p->d_status, sizeof(DevStatus), hipMemcpyDeviceToHost, ctx->stream));
    DRX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (total_words) *total_words = p->h_status->total_words;
    // what an encode measured decides the next encode's kernel (stream_encoder_suits())
    if (p->last_was_encode && !p->h_status->err && p->G.total_waves) p->enc_words_per_wave = p->h_status->total_words / p->G.total_waves;
    if (p->h_status->err & kErrInternal) return fail(ctx, DRX_ERR_DEVICE, "encoder look-back timed out (internal error)");
    if (p->h_status->err & kErrCorrupt) return fail(ctx, DRX_ERR_CORRUPT, "encoded input failed header-chain validation");
    if (p->h_status->err & kErrCapacity)
        return fail(ctx, DRX_ERR_CAPACITY, "encoded batch needs %llu words", (unsigned long long)p->h_status->total_words);
    return DRX_OK;
}

// ---------------------------------------------------------------------------
// one chunk through host memory: the body of the H5Z callback
// ---------------------------------------------------------------------------
static drx_status grow(drx_ctx *ctx, void **ptr, size_t *cap, size_t need, bool pinned) {
    if (*cap >= need) return DRX_OK;
    size_t want = need + need / 4 + 4096;
    if (*ptr) { if (pinned) (void)hipHostFree(*ptr); else (void)hipFree(*ptr); *ptr = nullptr; *cap = 0; }
    hipError_t e = pinned ? hipHostMalloc(ptr, want, hipHostMallocDefault) : hipMalloc(ptr, want);
    if (e != hipSuccess) return fail(ctx, DRX_ERR_DEVICE, "allocation of %zu bytes failed: %s", want, hipGetErrorString(e));
    *cap = want;
    return DRX_OK;
}

drx_status drx_filter_chunk_host(drx_ctx *ctx, int reverse, size_t cd_nelmts, const unsigned *cd_values,
                                 const void *in, size_t nbytes, void **out, size_t *out_bytes) {
    if (!ctx || !in || !out || !out_bytes) return DRX_ERR_ARG;
    drx_opts o;
    if (drx_parse_cd_values(cd_nelmts, cd_values, &o) != DRX_OK)
        return fail(ctx, DRX_ERR_ARG, "invalid compression_opts");
    std::lock_guard<std::mutex> lock(ctx->mu);
    DRX_ON_DEVICE(ctx);
    if (!ctx->d_off) DRX_HIP(ctx, hipMalloc((void **)&ctx->d_off, 2 * sizeof(uint64_t)));

    uint32_t n_samples;
    if (!reverse) {
        // nbytes must be a whole number of int16 (src/deltaRice.c:394-397) and < 2^31 samples (:389)
        if (nbytes == 0 || (nbytes & 1u) || nbytes / 2 > 0x7fffffffull) return fail(ctx, DRX_ERR_ARG, "bad chunk size %zu", nbytes);
        n_samples = (uint32_t)(nbytes / 2);
    } else {
        if (nbytes < 8 || (nbytes & 3u)) return fail(ctx, DRX_ERR_CORRUPT, "encoded chunk of %zu bytes", nbytes);
        memcpy(&n_samples, in, 4);  // totalNumberPoints (:306)
        if (n_samples == 0 || n_samples > 0x7fffffffu) return fail(ctx, DRX_ERR_CORRUPT, "bad sample count in header");
        // the header is untrusted and sizes every allocation below: a code has at least 1 + k bits (:215-222), so a
        // chunk of nbytes cannot hold more than 8 * (nbytes - 8) / (1 + k) samples
        if ((uint64_t)(nbytes - 8) * 8u < (uint64_t)n_samples * (o.rice_k + 1u))
            return fail(ctx, DRX_ERR_CORRUPT, "header claims %u samples, %zu bytes cannot hold them", n_samples, nbytes);
    }
    const uint32_t L = (o.wave_len < 0) ? 0u : (uint32_t)o.wave_len;
    drx_status st = DRX_OK;
    if (!ctx->host_plan || ctx->hp_samples != n_samples || ctx->hp_L != L || ctx->hp_k != o.rice_k ||
        ctx->hp_ntaps != o.n_taps || memcmp(ctx->hp_taps, o.taps, o.n_taps * sizeof(int32_t)) != 0) {
        if (ctx->host_plan) { plan_free(ctx->host_plan); ctx->host_plan = nullptr; }
        drx_plan *np = nullptr;
        if ((st = drx_plan_create_uniform(ctx, 1, n_samples, L, o.rice_k, &np)) != DRX_OK) return st;
        if ((st = drx_plan_set_filter(np, o.n_taps, o.taps)) != DRX_OK) { plan_free(np); return st; }
        ctx->host_plan = np;
        ctx->hp_samples = n_samples; ctx->hp_L = L; ctx->hp_k = o.rice_k; ctx->hp_ntaps = o.n_taps;
        memcpy(ctx->hp_taps, o.taps, sizeof ctx->hp_taps);
    }
    drx_plan *plan = ctx->host_plan;
    // DRX_TRACE=1: wall time of each phase of the call on stderr
    static const bool trace = getenv("DRX_TRACE") != nullptr;
    auto now = []() { return std::chrono::steady_clock::now(); };
    auto t_prev = now();
    auto lap = [&](const char *what) {
        if (!trace) return;
        const auto t = now();
        fprintf(stderr, "[drx] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t - t_prev).count());
        t_prev = t;
    };
    const size_t raw_bytes = (size_t)n_samples * 2;
    const size_t enc_cap_bytes = (size_t)plan->max_words * 4;
    void *result = nullptr;
    do {
        if ((st = grow(ctx, &ctx->d_raw, &ctx->raw_cap, raw_bytes, false)) != DRX_OK) break;
        if ((st = grow(ctx, &ctx->d_enc, &ctx->enc_cap, reverse ? nbytes : enc_cap_bytes, false)) != DRX_OK) break;
        hipError_t e;
        if (!reverse) {
            e = hipMemcpyAsync(ctx->d_raw, in, raw_bytes, hipMemcpyHostToDevice, ctx->stream);
            if (e != hipSuccess) { st = fail(ctx, DRX_ERR_DEVICE, "H2D failed: %s", hipGetErrorString(e)); break; }
            lap("encode: H2D issue");
            if ((st = drx_encode(plan, (const int16_t *)ctx->d_raw, (uint32_t *)ctx->d_enc, plan->max_words, ctx->d_off)) != DRX_OK) break;
            uint64_t words = 0;
            if ((st = drx_plan_finish(plan, &words)) != DRX_OK) break;
            lap("encode: kernels + finish");
            const size_t nb = (size_t)words * 4;
            result = malloc(nb);
            if (!result) { st = DRX_ERR_NOMEM; break; }
            e = hipMemcpyAsync(result, ctx->d_enc, nb, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            lap("encode: D2H");
            if (e != hipSuccess) { st = fail(ctx, DRX_ERR_DEVICE, "D2H failed: %s", hipGetErrorString(e)); break; }
            *out_bytes = nb;
        } else {
            // The device finds the headers of a large chunk (parallel walk: 0.14 ms for 2000 waveforms; the CPU takes 0.28 ms,
            // DRX_HOST_WALK=1 forces that, =0 forbids it).
            const uint64_t W = plan->G.total_waves;
            // A chunk of a few hundred waveforms is walked on the CPU as well: ~0.14 us per hop out of host memory against
            // ~1 us per dependent load on the device (20 waveforms: 19 us) or the 60 us of the parallel walk's three launches.
            static const char *walk_env = getenv("DRX_HOST_WALK");
            const bool device_walk = walk_env ? atoi(walk_env) == 0 : W > 512u;
            const size_t tab_bytes = 16 + (device_walk ? 0 : (size_t)W * (sizeof(uint64_t) + sizeof(uint32_t)));
            if ((st = grow(ctx, &ctx->h_pin, &ctx->pin_cap, tab_bytes, true)) != DRX_OK) break;
            uint64_t *h_off = (uint64_t *)ctx->h_pin;              // [2] chunk table, then [W] wave_off
            uint32_t *h_words = (uint32_t *)(h_off + 2 + W);       // [W] wave_words
            h_off[0] = 0;
            h_off[1] = (uint64_t)(nbytes / 4);
            e = hipMemcpyAsync(ctx->d_enc, in, nbytes, hipMemcpyHostToDevice, ctx->stream);
            lap("decode: H2D chunk");
            if (e == hipSuccess) e = hipMemcpyAsync(ctx->d_off, h_off, 2 * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream);
            if (!device_walk) {
                if (!walk_chunk_host((const uint32_t *)in, nbytes / 4, n_samples, L, o.rice_k, h_off + 2, h_words)) {
                    (void)hipStreamSynchronize(ctx->stream);
                    st = fail(ctx, DRX_ERR_CORRUPT, "encoded chunk failed header-chain validation");
                    break;
                }
                lap("decode: host walk");
                if (e == hipSuccess) e = hipMemcpyAsync(plan->d_wave_off, h_off + 2, W * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream);
                if (e == hipSuccess) e = hipMemcpyAsync(plan->d_wave_words, h_words, W * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream);
                lap("decode: H2D tables");
            }
            // (no wait here: the pinned table is not touched again before this call's final synchronisation)
            if (e != hipSuccess) { st = fail(ctx, DRX_ERR_DEVICE, "H2D failed: %s", hipGetErrorString(e)); break; }
            if ((st = decode_launch(plan, (const uint32_t *)ctx->d_enc, nbytes / 4, ctx->d_off, (int16_t *)ctx->d_raw, !device_walk)) != DRX_OK) break;
            if ((st = drx_plan_finish(plan, nullptr)) != DRX_OK) break;
            lap("decode: kernels + finish");
            result = malloc(raw_bytes);
            if (!result) { st = DRX_ERR_NOMEM; break; }
            e = hipMemcpyAsync(result, ctx->d_raw, raw_bytes, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            lap("decode: D2H");
            if (e != hipSuccess) { st = fail(ctx, DRX_ERR_DEVICE, "D2H failed: %s", hipGetErrorString(e)); break; }
            *out_bytes = raw_bytes;
        }
    } while (0);
    if (st != DRX_OK) { free(result); return st; }
    *out = result;
    return DRX_OK;
}

}  // extern "C"
