// drx_pieces.hip -- single-pass encoder for batches whose waveforms are much shorter or much longer than the
// ~2000-8000 samples one wavefront of k_encode_fused (drx_encode_kernels.hip) likes (gfx950).
//
// k_encode_fused gives a waveform to a wavefront: with WaveformLength 512 it pays a workgroup barrier, a share of a
// look-back and an 8 KB LDS clear per 512 samples, and beyond ~10 000 samples the code outgrows the wavefront's LDS
// buffer and the waveform is encoded twice.  Those batches went to the two-pass segment encoder (sizes, scan, pack: every
// sample read and coded twice).  Here the unit a wavefront takes is a PIECE of a chunk:
//   run      WaveformLength <= 3584: kPcRunSamples / L consecutive whole waveforms.  Their streams are word aligned and
//            follow one another in the encoded chunk (src/deltaRice.c:427-433), so the wavefront builds
//            [n_0 | payload_0 | n_1 | payload_1 | ...] in its LDS buffer and copies it out in one piece;
//   segment  WaveformLength > 10 240: the waveform is cut into 2, 4 or 8 segments of whole tiles, one wavefront each, all in
//            ONE workgroup.  A segment is coded from bit 0 of its wavefront's buffer; once the workgroup knows the bits
//            of every segment, segment s is copied out shifted to its bit position inside the waveform's stream, the word
//            it shares with segment s + 1 completed from that wavefront's buffer (LDS, no global atomics, no zeroing);
//   else     one waveform, as k_encode_fused.
// PACKED (every WaveformLength below 512 and a multiple of 8): a 512-sample tile per waveform would be mostly empty lanes, so
// a run's samples form ONE sequence of full tiles that span waveform boundaries (a lane's 8 samples belong to one
// waveform); segmented scans give every lane its bit inside its waveform and every waveform its header word.
// GEN: a forward filter of up to four taps (src/deltaRice.c:64-74) instead of the delta; its three samples of history cross
// tile, segment and part boundaries and stop at waveform boundaries.
// SUPER (every WaveformLength above 65 536, among them the reference's default of one waveform per chunk): a waveform is
// cut into PARTS of eight segments, a workgroup each.  The bits in front of a part come from a second look-back over the
// parts of its waveform, the waveform's place from the first one, which then has one entry per WAVEFORM, published by
// the waveform's last part.  The word a part shares with the next one is completed by its last wavefront, which codes
// the 32 samples that follow its segment as well (at least 32 bits) -- so the parts exchange nothing but their bit counts.
// The pieces of a chunk fill whole workgroups (a workgroup never spans two chunks), one ticket and one look-back entry
// per workgroup as in k_encode_fused.  A piece whose code outgrows the 8 KB buffer (incompressible data) is coded again
// after the look-back, tile by tile to its final position, by one wavefront per waveform.
//
// The arithmetic of a tile (packed 16-bit code lengths, DPP scan, lane-local concatenation, ds_or) is drx_encode.h, the
// same code k_encode_fused runs; reference: src/deltaRice.c:49-63 (delta), :191-244 (Rice code), :383-441 (framing).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "drx_internal.h"
#include "drx_device.h"
#include "drx_encode.h"

namespace drx {

namespace {

struct PcChunk {
    uint64_t c;           // chunk
    uint64_t sample_off;  // its first sample
    uint64_t wave_base;   // its first waveform
    uint32_t n_samples, L, W;
    uint32_t j;  // workgroup inside the chunk
};

__device__ __forceinline__ uint32_t rfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t rfl64(uint64_t v) { return ((uint64_t)rfl((uint32_t)(v >> 32)) << 32) | rfl((uint32_t)v); }

// workgroup (ticket) T -> chunk
__device__ __forceinline__ PcChunk pc_locate(const Geom &G, uint32_t T, bool packed) {
    PcChunk q;
    if (G.uniform) {
        const uint32_t wgs = piece_shape(G.u_wave_len, G.u_n_waves, G.k, packed).wgs;
        q.c = T / wgs;
        q.j = T - (uint32_t)q.c * wgs;
        q.sample_off = q.c * (uint64_t)G.u_n_samples;
        q.wave_base = q.c * (uint64_t)G.u_n_waves;
        q.n_samples = G.u_n_samples;
        q.L = G.u_wave_len;
        q.W = G.u_n_waves;
    } else {
        uint64_t lo = 0, hi = G.n_chunks;  // invariant: pc_wg_base[lo] <= T < pc_wg_base[hi]
        while (hi - lo > 1) {
            const uint64_t mid = (lo + hi) >> 1;
            if (G.pc_wg_base[mid] <= T) lo = mid; else hi = mid;
        }
        const ChunkDesc d = G.chunks[lo];
        q.c = lo;
        q.j = T - G.pc_wg_base[lo];
        q.sample_off = d.sample_off;
        q.wave_base = d.wave_base;
        q.n_samples = d.n_samples;
        q.L = d.wave_len;
        q.W = d.n_waves;
    }
    return q;
}

// Decoupled look-back, the reading half: sum of the values of entries [first, idx), 128 entries per poll.  Owners publish
// kScanAgg | value, later kScanPrefix | sum of [first, own]; a zero entry is not there yet.  One wavefront calls this.
__device__ __forceinline__ uint64_t lookback_sum(const uint64_t *state, int64_t idx, int64_t first, int lane, DevStatus *st) {
    uint64_t sum = 0;
    int64_t base = idx - 1;
    uint32_t spins = 0;
    // the nearest predecessor alone first, one 8-byte load per poll: hundreds of waiting workgroups polling whole windows
    // were a fabric load of their own (k_encode_fused: 5.86 -> 5.43 ms, profiles/r03_notes.md)
    while (base >= first) {
        uint64_t v = 0;
        if (lane == 0) v = __hip_atomic_load(state + base, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 62)) != 0) break;
        __builtin_amdgcn_s_sleep(8);
        if (++spins > (1u << 22)) break;  // (the window loop below reports it)
    }
    for (;;) {
        const int64_t i0 = base - lane, i1 = base - 64 - lane;
        uint64_t s0v = kScanPrefix, s1v = kScanPrefix;  // in front of `first`: an empty prefix
        if (i0 >= first) s0v = __hip_atomic_load(state + i0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (i1 >= first) s1v = __hip_atomic_load(state + i1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t st0 = (uint32_t)(s0v >> 62), st1 = (uint32_t)(s1v >> 62);
        const uint64_t p0 = __ballot(st0 == 2u), z0 = __ballot(st0 == 0u);
        const uint64_t p1 = __ballot(st1 == 2u), z1 = __ballot(st1 == 0u);
        const int fp = p0 ? __builtin_ctzll(p0) : (p1 ? 64 + __builtin_ctzll(p1) : 128);
        const uint64_t near0 = fp >= 64 ? ~0ull : ((1ull << fp) - 1ull);
        const uint64_t near1 = fp >= 128 ? ~0ull : (fp > 64 ? ((1ull << (fp - 64)) - 1ull) : 0ull);
        if ((z0 & near0) | (z1 & near1)) {  // a nearer predecessor has not published yet
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 22)) {  // cannot happen with zeroed state (ticket holders' predecessors run); never hang the GPU
                if (lane == 0) atomicOr(&st->err, kErrInternal);
                break;
            }
            continue;
        }
        const uint64_t c0 = (lane <= fp) ? (s0v & kScanValMask) : 0ull;
        const uint64_t c1 = (64 + lane <= fp) ? (s1v & kScanValMask) : 0ull;
        sum += wave_sum_u64(c0 + c1);
        if (fp < 128) break;
        base -= 128;
    }
    return sum;
}

__device__ __forceinline__ void scan_publish(uint64_t *state, int64_t idx, uint64_t tagged, int lane) {
    if (lane == 0) __hip_atomic_store(state + idx, tagged, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace

template <bool SUPER, bool GEN, bool PACKED>
__global__ __launch_bounds__(64 * kPcWaves) void k_encode_pieces(Geom G, const int16_t *__restrict__ in, uint64_t in_samples,
                                                                 uint32_t *__restrict__ out, uint64_t out_cap,
                                                                 uint64_t *__restrict__ chunk_word_off,
                                                                 uint32_t *__restrict__ wave_words,
                                                                 uint64_t *__restrict__ scan_state, uint64_t *__restrict__ part_state,
                                                                 uint32_t *__restrict__ ticket, uint32_t total_wgs, DevStatus *st) {
    // per wavefront: 4 pad words (place_words ORs zeros below a lane's first word), the code, 4 slack words
    __shared__ __attribute__((aligned(16))) uint32_t buf_all[kPcWaves][kEncCapWords + 8];
    __shared__ uint32_t s_ticket;
    __shared__ uint32_t s_n[kPcWaves][kPcMaxRun];  // runs: n_i of the run's waveforms
    __shared__ uint32_t s_size[kPcWaves];          // runs: words of the run, headers included; segments: bits of the segment
    __shared__ uint32_t s_fit[kPcWaves];           // the piece is in its buffer
    __shared__ uint64_t s_excl, s_excl_bits, s_wf_words;
    const int lane = lane_id();
    const uint32_t wv = rfl(threadIdx.x >> 6);
    uint32_t *row = buf_all[wv];
    uint32_t *buf = row + 4;
    const uint32_t buf_bits = lds_addr(buf) * 8u;

    if (threadIdx.x == 0) s_ticket = atomicAdd(ticket, 1u);
    for (int i = lane; i < (int)(kEncCapWords + 8) / 4; i += 64) reinterpret_cast<uint4 *>(row)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    const uint32_t T = s_ticket;
    if (T >= total_wgs) return;  // (the grid is exactly total_wgs workgroups)
    PcChunk q = pc_locate(G, T, PACKED);
    q.n_samples = rfl(q.n_samples); q.L = rfl(q.L); q.W = rfl(q.W); q.j = rfl(q.j);
    q.sample_off = rfl64(q.sample_off); q.wave_base = rfl64(q.wave_base); q.c = rfl64(q.c);
    const PieceShape sh = piece_shape(q.L, q.W, G.k, PACKED);
    const uint32_t p = rfl(q.j * kPcWaves + wv);  // piece of the chunk
    const bool live = SUPER || p < sh.pieces;
    const bool runs = !SUPER && (PACKED || sh.run > 1u);
    const uint32_t S = sh.segs;
    const uint32_t k = G.k;
    // SUPER: workgroup j of the chunk is part `part` of waveform w0, this wavefront its segment part * kPcWaves + wv
    const uint32_t parts = SUPER ? sh.parts : 1u;
    const uint32_t part = SUPER ? q.j % parts : 0u;

    // what this wavefront encodes: `nspans` spans of samples, contiguous in memory from x on
    //   runs:     span i = waveform w0 + i of the chunk (whole);        segments: one span = segment sg of waveform w0
    //   SUPER:    a part's last wavefront takes a second span, the (up to) 32 samples behind its segment
    const uint32_t w0 = SUPER ? q.j / parts : (runs ? p * sh.run : p / S);
    const uint32_t sg = SUPER ? part * kPcWaves + wv : (runs ? 0u : p % S);
    uint32_t nspans = 0;
    uint32_t wf_len = 0;  // segments: samples of the whole waveform
    uint32_t extra = 0;   // SUPER: samples of the second span
    uint32_t run_wfs = 0, run_samples = 0;  // PACKED: waveforms and samples of the run
    if (live) {
        if (runs) {
            nspans = q.W - w0 < sh.run ? q.W - w0 : sh.run;
            if (PACKED) {  // the run's samples as ONE span: tiles span waveform boundaries
                run_wfs = nspans;
                run_samples = (w0 + nspans == q.W) ? q.n_samples - w0 * q.L : nspans * q.L;
                nspans = 1u;
            }
        } else {
            wf_len = (w0 + 1u == q.W) ? q.n_samples - w0 * q.L : q.L;
            nspans = (uint64_t)sg * sh.seg_len < wf_len ? 1u : 0u;
            if (SUPER && nspans && wv + 1u == kPcWaves && (uint64_t)(sg + 1u) * sh.seg_len < wf_len) {
                const uint32_t left = wf_len - (sg + 1u) * sh.seg_len;
                extra = left < 32u ? left : 32u;
                nspans = 2u;
            }
        }
    }
    const uint64_t wf_off = q.sample_off + (uint64_t)w0 * q.L;  // first sample of waveform w0 in the batch
    const uint64_t xoff = wf_off + (uint64_t)sg * sh.seg_len;   // first sample of this piece
    auto span_len = [&](uint32_t i) -> uint32_t {
        if (PACKED) return run_samples;
        if (runs) return (w0 + i + 1u == q.W) ? q.n_samples - (w0 + i) * q.L : q.L;
        if (SUPER && i == 1u) return extra;
        const uint32_t left = wf_len - sg * sh.seg_len;
        return left < sh.seg_len ? left : sh.seg_len;
    };
    // the sample in front of a segment (a waveform's first sample has none: x[-1] := 0, src/deltaRice.c:53-54)
    // (GEN, a forward filter of up to four taps, src/deltaRice.c:64-74: the three samples in front, as two dwords)
    uint32_t carry0 = 0, carry2_0 = 0;
    if (!runs && sg && nspans) {
        carry0 = (uint32_t)(uint16_t)in[xoff - 1u] << 16;
        if (GEN) {  // (a segment starts at least 512 samples into its waveform)
            carry0 |= (uint32_t)(uint16_t)in[xoff - 2u];
            carry2_0 = (uint32_t)(uint16_t)in[xoff - 4u] | ((uint32_t)(uint16_t)in[xoff - 3u] << 16);
        }
    }
    const u16x2 tp[4] = {splat(GEN ? G.enc_t[0] : 1u), splat(GEN ? G.enc_t[1] : 0xffffu), splat(GEN ? G.enc_t[2] : 0u),
                         splat(GEN ? G.enc_t[3] : 0u)};

    // ---- the tiles of all spans as one sequence: loads run kDepth tiles ahead, across span boundaries ----
    struct Cursor { uint32_t i, off, rem; };  // span, sample offset of the tile from xoff, samples of the span from this tile on
    auto cur_first = [&]() -> Cursor { Cursor c; c.i = 0; c.off = 0; c.rem = nspans ? span_len(0) : 0u; return c; };
    auto cur_next = [&](Cursor &c) {
        if (c.i >= nspans) return;
        if (c.rem > (uint32_t)kTile) { c.off += (uint32_t)kTile; c.rem -= (uint32_t)kTile; return; }
        c.off += c.rem;  // (runs: the next waveform follows in memory)
        c.i += 1u;
        c.rem = c.i < nspans ? span_len(c.i) : 0u;
    };
    // branch-free 16-byte load at any int16 alignment, clamped to the batch: a lane whose 8 samples would pass the end of
    // the input reads the last 8 samples instead and shifts them into place when the tile is consumed (fix_tail)
    const uint64_t last8 = in_samples - 8u;  // (pieces_batch(): the batch has at least one tile of samples)
    auto load_item = [&](const Cursor &c) -> uint4 {
        const uint64_t gi = xoff + (c.i < nspans ? c.off : 0u) + 8u * (uint32_t)lane;
        const uint64_t a = gi < last8 ? gi : last8;
        return *reinterpret_cast<const uint4 *>(in + a);
    };
    auto fix_tail = [&](const Cursor &c, uint32_t (&w)[4]) {
        const uint64_t gi = xoff + c.off + 8u * (uint32_t)lane;
        if (gi > last8) {  // rare: the last lanes of the batch's last tile
            const uint32_t d = (uint32_t)(gi - last8);  // samples to drop from the front (1..7 matter; more: nothing valid)
            for (uint32_t s = 0; s < d && s < 8u; ++s) {
                w[0] = __builtin_amdgcn_alignbit(w[1], w[0], 16);
                w[1] = __builtin_amdgcn_alignbit(w[2], w[1], 16);
                w[2] = __builtin_amdgcn_alignbit(w[3], w[2], 16);
                w[3] >>= 16;
            }
        }
    };

    uint32_t wpos = 0;   // runs: word of buf where the current waveform's header goes
    uint32_t Pw = 0;     // bits of the current span so far (SUPER: of both spans)
    uint32_t seg_bits = 0;  // SUPER: bits of the segment itself
    uint32_t open_bits = 0;  // PACKED: bits so far of the waveform that is open where the tile begins (its header word: wpos)
    uint32_t pk_wi = 0;
    const uint32_t pk_magic = PACKED ? 0xffffffffu / q.L + 1u : 0u;
    bool fits = true;    // everything so far is in buf
    uint32_t carry = 0;  // dword whose high half is the sample before the tile
    uint32_t carry2 = 0;  // GEN: the dword before that one
    auto process = [&](const Cursor &c, const uint4 &qv, auto full_tag) __attribute__((always_inline)) {
        constexpr bool FULLT = decltype(full_tag)::value;
        uint32_t w[4] = {qv.x, qv.y, qv.z, qv.w};
        if (!FULLT) fix_tail(c, w);
        uint32_t xprev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w[3], 0x138, 0xf, 0xf, false);  // wave_shr:1
        if (lane == 0) xprev = carry;
        carry = (uint32_t)__builtin_amdgcn_readlane((int)w[3], 63);
        uint32_t xprev2 = 0;
        if (GEN) {
            xprev2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w[2], 0x138, 0xf, 0xf, false);  // wave_shr:1
            if (lane == 0) xprev2 = carry2;
            carry2 = (uint32_t)__builtin_amdgcn_readlane((int)w[2], 63);
        }
        // PACKED: which waveform of the run my 8 samples belong to (WaveformLength is a multiple of 8: all to one), whether
        // they open or close it
        uint32_t pk_g0 = 0, pk_wstart = 0;
        bool pk_end = false, pk_in = false;
        if (PACKED) {
            pk_g0 = c.off + 8u * (uint32_t)lane;          // my first sample, counted from the run's
            const uint32_t wi = __umulhi(pk_g0, pk_magic);  // = pk_g0 / L (exact: pk_g0 L < 2^32)
            pk_wstart = wi * q.L;
            pk_in = pk_g0 < run_samples;
            pk_end = pk_in && (pk_g0 + 8u >= pk_wstart + q.L || pk_g0 + 8u >= run_samples);
            if (pk_g0 == pk_wstart) { xprev = 0; xprev2 = 0; }  // a waveform starts from x[-1] := 0 (src/deltaRice.c:53-54)
            pk_wi = wi;
        }
        PackedCodes pc;
        packed_codes<GEN>(w, xprev, xprev2, tp, k, pc);
        if (!FULLT) {
            const uint32_t l8 = 8u * (uint32_t)lane;
            const int nv = c.rem <= l8 ? 0 : (int)(c.rem - l8 < 8u ? c.rem - l8 : 8u);
            mask_tail(pc, nv);
        }
        const uint32_t lane_bits = lane_tile_bits(pc);
        uint32_t cw[4];
        if (FULLT) concat_codes(pc, cw);
        const uint32_t incl = wave_incl_scan_dpp(lane_bits);
        const uint32_t tile_bits = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        if (PACKED) {
            // bits of my waveform in front of me: those of this tile (a segmented scan through the lane that opens the
            // waveform in this tile) + those of earlier tiles if it was open when the tile began
            const uint32_t ex = incl - lane_bits;
            const uint32_t s0 = pk_wstart <= c.off ? 0u : (pk_wstart - c.off) >> 3;  // first lane of my waveform in this tile
            const uint32_t ex_s = (uint32_t)__shfl((int)ex, (int)s0);
            const uint32_t my_bit = (pk_wstart < c.off ? open_bits : 0u) + ex - ex_s;
            const uint32_t n_w = (my_bit + lane_bits + 31u) >> 5;  // (of a lane that closes its waveform: n_i)
            const uint32_t wscan = wave_incl_scan_dpp(pk_end ? 1u + n_w : 0u);
            // words of the waveforms closed before mine (the shuffle outside any branch: it reads active lanes only)
            const uint32_t wprev = (uint32_t)__shfl((int)wscan, (int)((s0 - 1u) & 63u));
            const uint32_t before = s0 ? wprev : 0u;
            const uint32_t wpos_my = wpos + before;  // my waveform's header word
            // (a lane behind the run's end codes nothing; its position only has to lie inside the buffer)
            const uint32_t P_lane = pk_in ? 32u * (wpos_my + 1u) + my_bit : 32u * (wpos + 1u);
            const uint32_t done_words = (uint32_t)__builtin_amdgcn_readlane((int)wscan, 63);
            const uint32_t open_new = (uint32_t)__builtin_amdgcn_readlane((int)((pk_end || !pk_in) ? 0u : my_bit + lane_bits), 63);
            if (fits && ((32u * (wpos + done_words + 1u) + open_new + 31u) >> 5) < kEncCapWords) {
                if (FULLT && !__any(lane_bits > 128u))
                    place_words(cw, buf_bits + P_lane + lane_bits);
                else
                    emit_tile<FULLT>(pc, buf_bits + P_lane);
                if (pk_end) buf[wpos_my] = n_w;  // (after the tile's ORs, in program order: an OR of zero may touch this word)
            } else {
                fits = false;
            }
            if (pk_end) wave_words[q.wave_base + w0 + pk_wi] = n_w;
            wpos += done_words;
            open_bits = open_new;
            return;
        }
        const uint32_t P = (runs ? 32u * (wpos + 1u) : 0u) + Pw;  // bit of buf where the tile starts
        if (fits && ((P + tile_bits + 31u) >> 5) < kEncCapWords) {
            if (FULLT && !__any(lane_bits > 128u))
                place_words(cw, buf_bits + P + incl);
            else
                emit_tile<FULLT>(pc, buf_bits + P + incl - lane_bits);
        } else {
            fits = false;
        }
        Pw += tile_bits;
    };
    auto consume = [&](Cursor &c, const uint4 &qv) __attribute__((always_inline)) {
        if (c.i >= nspans) return;
        if (c.rem >= (uint32_t)kTile) process(c, qv, std::true_type{}); else process(c, qv, std::false_type{});
        if (c.rem <= (uint32_t)kTile) {  // the span's last tile
            if (runs && !PACKED) {
                const uint32_t n = (Pw + 31u) >> 5;
                if (lane == 0) {
                    if (fits) buf[wpos] = n;
                    s_n[wv][c.i] = n;
                }
                wpos += 1u + n;
                Pw = 0;
                carry = 0;  // the next waveform starts from x[-1] := 0
                carry2 = 0;
            }
            if (SUPER && c.i == 0u) seg_bits = Pw;  // (what follows belongs to the next part)
        }
        cur_next(c);
    };
    {
        constexpr int kDepth = 3;
        carry = carry0;
        carry2 = carry2_0;
        Cursor pcur = cur_first(), lcur = pcur;
        uint4 qv[kDepth];
#pragma unroll
        for (int u = 0; u < kDepth; ++u) { qv[u] = load_item(lcur); cur_next(lcur); }
#pragma unroll 1
        while (pcur.i < nspans) {
#pragma unroll
            for (int u = 0; u < kDepth; ++u) {
                consume(pcur, qv[u]);
                qv[u] = load_item(lcur);
                cur_next(lcur);
            }
        }
    }
    wave_sync();
    if (lane == 0) {
        s_size[wv] = runs ? wpos : (SUPER ? seg_bits : Pw);
        s_fit[wv] = fits ? 1u : 0u;
    }
    __syncthreads();

    // ---- sizes of this workgroup's waveforms, look-back ----
    const bool first_wg = q.j == 0u;
    // segments: the group of S wavefronts of my waveform, its bits before my segment, its words, the words of the
    // groups in front of it, and whether every segment of the group is in its buffer
    const uint32_t gb = runs ? wv : (wv & ~(S - 1u));
    uint32_t bits_before = 0, grp_bits = 0, words_before = 0, block_words = 0;
    bool grp_fit = true;
    if (runs) {
#pragma unroll
        for (uint32_t i = 0; i < kPcWaves; ++i) {
            const uint32_t sz = s_size[i];
            words_before += i < wv ? sz : 0u;
            block_words += sz;
        }
        grp_fit = fits;
    } else {
#pragma unroll
        for (uint32_t g0 = 0; g0 < kPcWaves; g0 += 1u) {
            if ((g0 & (S - 1u)) != 0u) continue;  // first wavefront of a group
            uint32_t gbits = 0;
            bool gf = true;
            for (uint32_t s = 0; s < S; ++s) {
                const uint32_t b = s_size[g0 + s];
                if (g0 == gb && s < sg) bits_before += b;
                gbits += b;
                gf = gf && s_fit[g0 + s];
            }
            const bool glive = (q.j * kPcWaves + g0) < sh.pieces;
            const uint32_t gw = glive ? 1u + ((gbits + 31u) >> 5) : 0u;
            if (g0 < gb) words_before += gw;
            if (g0 == gb) { grp_bits = gbits; grp_fit = gf; }
            block_words += gw;
        }
    }
    const uint64_t block_sum = (uint64_t)block_words + (first_wg ? 1ull : 0ull);
    // SUPER: the part's bits and whether all of it is in LDS
    uint32_t part_bits = 0;
    bool part_fit = true;
    if (SUPER) {
#pragma unroll
        for (uint32_t i = 0; i < kPcWaves; ++i) { part_bits += s_size[i]; part_fit = part_fit && s_fit[i]; }
    }
    const bool last_part = part + 1u == parts;
    const bool chunk_first = SUPER ? (w0 == 0u && part == 0u) : first_wg;
    if (wv == 0) {
        uint64_t excl_words = 0, sum_words = block_sum;
        if (!SUPER) {
            // decoupled look-back over one entry per workgroup (as in k_encode_fused)
            if (T == 0) {
                scan_publish(scan_state, T, kScanPrefix | block_sum, lane);
            } else {
#ifdef DRX_PC_NOLB
                excl_words = (uint64_t)T * 12000ull;  // ablation: no look-back (positions wrong, results invalid)
#else
                scan_publish(scan_state, T, kScanAgg | block_sum, lane);
                excl_words = lookback_sum(scan_state, T, 0, lane, st);
                scan_publish(scan_state, T, kScanPrefix | (excl_words + block_sum), lane);
#endif
            }
        } else {
            // bits of the waveform in front of this part: look-back over the waveform's parts (workgroups T - part .. T)
            uint64_t excl_bits = 0;
            if (part == 0u) {
                scan_publish(part_state, T, kScanPrefix | (uint64_t)part_bits, lane);
            } else {
                scan_publish(part_state, T, kScanAgg | (uint64_t)part_bits, lane);
                excl_bits = lookback_sum(part_state, T, (int64_t)T - (int64_t)part, lane, st);
                scan_publish(part_state, T, kScanPrefix | (excl_bits + part_bits), lane);
            }
            // words in front of the waveform: look-back over one entry per WAVEFORM, which its last part publishes
            const int64_t U = (int64_t)(q.wave_base + w0);
            const uint64_t n_total = (excl_bits + part_bits + 31u) >> 5;  // (last part: the waveform's payload words)
            sum_words = 1ull + n_total + (w0 == 0u ? 1ull : 0ull);
            if (last_part) {
                if (U == 0) {
                    scan_publish(scan_state, U, kScanPrefix | sum_words, lane);
                } else {
                    scan_publish(scan_state, U, kScanAgg | sum_words, lane);
                    excl_words = lookback_sum(scan_state, U, 0, lane, st);
                    scan_publish(scan_state, U, kScanPrefix | (excl_words + sum_words), lane);
                }
            } else if (U != 0) {
                excl_words = lookback_sum(scan_state, U, 0, lane, st);
            }
            if (lane == 0) { s_excl_bits = excl_bits; s_wf_words = n_total; }
        }
        if (lane == 0) {
            s_excl = excl_words;
            if (chunk_first) {
                chunk_word_off[q.c] = excl_words;
                if (excl_words < out_cap) out[excl_words] = q.n_samples;  // chunk header, src/deltaRice.c:415
            }
            if (T + 1u == total_wgs) {  // (SUPER: the last part of the last waveform)
                chunk_word_off[G.n_chunks] = excl_words + sum_words;
                st->total_words = excl_words + sum_words;
                if (G.host_words) *G.host_words = excl_words + sum_words;
                if (excl_words + sum_words > out_cap) atomicOr(&st->err, kErrCapacity);
            }
        }
    }
    __syncthreads();
    if (!live) return;
    // runs: first header of the run; segments: the waveform's header
    const uint64_t pos = s_excl + ((SUPER ? w0 == 0u : first_wg) ? 1ull : 0ull) + words_before;

    // a waveform coded once more, tile by tile, straight to its place (its code did not fit the buffer)
    auto stream_waveform = [&](uint64_t soff, uint32_t len, uint32_t *__restrict__ outp) -> uint64_t {
        for (int i = lane; i < (int)(kEncCapWords + 8) / 4; i += 64) reinterpret_cast<uint4 *>(row)[i] = make_uint4(0, 0, 0, 0);
        wave_sync();
        const int16_t *x = in + soff;
        uint64_t P = 0;
        uint32_t cr = 0, cr2 = 0;
        uint32_t wn[4];
        int nvn = load8_dwords(x, len, 0u, lane, true, wn);
        for (uint32_t t0 = 0; t0 < len; t0 += kTile) {
            uint32_t w[4] = {wn[0], wn[1], wn[2], wn[3]};
            const int nv = nvn;
            if (t0 + kTile < len) nvn = load8_dwords(x, len, t0 + kTile, lane, true, wn);  // (travels while this tile is coded)
            uint32_t xprev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w[3], 0x138, 0xf, 0xf, false);
            if (lane == 0) xprev = cr;
            cr = (uint32_t)__shfl((int)w[3], 63);
            uint32_t xprev2 = 0;
            if (GEN) {
                xprev2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w[2], 0x138, 0xf, 0xf, false);
                if (lane == 0) xprev2 = cr2;
                cr2 = (uint32_t)__shfl((int)w[2], 63);
            }
            PackedCodes pc;
            packed_codes<GEN>(w, xprev, xprev2, tp, k, pc);
            mask_tail(pc, nv);
            const uint32_t lane_bits = lane_tile_bits(pc);
            const uint32_t incl = wave_incl_scan_dpp(lane_bits);
            const uint32_t tile_bits = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            const uint64_t wfirst = P >> 5;  // first staged word
            emit_tile<false>(pc, buf_bits + (uint32_t)(P & 31u) + incl - lane_bits);
            P += tile_bits;
            wave_sync();
            const uint32_t nfull = (uint32_t)((P >> 5) - wfirst);
            for (uint32_t i = lane; i < nfull; i += 64) { outp[wfirst + i] = buf[i]; buf[i] = 0; }
            wave_sync();
            if (nfull && lane == 0) { const uint32_t cwd = buf[nfull]; buf[nfull] = 0; buf[0] = cwd; }
            wave_sync();
        }
        if ((P & 31u) && lane == 0) outp[P >> 5] = buf[0];
        return P;
    };

    // Samples [s_begin, s_end) of the waveform once more, tile by tile, from bit Bp of the waveform's stream on (their code did
    // not fit the buffer).  The word in which they start belongs to whoever codes the samples in front; the word in which they
    // end is completed from the (up to) 32 samples that follow (at least a bit each): wavefronts exchange nothing.  The next
    // tile's samples travel while this one is coded.
    auto stream_samples = [&](uint32_t s_begin, uint32_t s_end, uint64_t Bp, uint32_t bits_mine, uint32_t *__restrict__ outp,
                              uint64_t cap_words) {
        const uint32_t P0 = (uint32_t)(Bp & 31u);
        const uint64_t wbase = Bp >> 5;
        const uint32_t limit = (P0 + bits_mine + 31u) >> 5, skip = P0 ? 1u : 0u;  // words [skip, limit) from wbase are mine
        const uint32_t more = wf_len - s_end < 32u ? wf_len - s_end : 32u;
        const uint32_t len = s_end - s_begin + more;
        for (int i = lane; i < (int)(kEncCapWords + 8) / 4; i += 64) reinterpret_cast<uint4 *>(row)[i] = make_uint4(0, 0, 0, 0);
        wave_sync();
        const int16_t *x = in + wf_off + s_begin;
        uint64_t P = P0;
        uint32_t cr = 0, cr2 = 0;
        if (s_begin) {
            cr = (uint32_t)(uint16_t)x[-1] << 16;
            if (GEN) {
                cr |= (uint32_t)(uint16_t)x[-2];
                cr2 = (uint32_t)(uint16_t)x[-4] | ((uint32_t)(uint16_t)x[-3] << 16);
            }
        }
        uint32_t wn[4];
        int nvn = load8_dwords(x, len, 0u, lane, true, wn);
        for (uint32_t t0 = 0; t0 < len; t0 += kTile) {
            uint32_t w[4] = {wn[0], wn[1], wn[2], wn[3]};
            const int nv = nvn;
            if (t0 + kTile < len) nvn = load8_dwords(x, len, t0 + kTile, lane, true, wn);
            uint32_t xprev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w[3], 0x138, 0xf, 0xf, false);
            if (lane == 0) xprev = cr;
            cr = (uint32_t)__shfl((int)w[3], 63);
            uint32_t xprev2 = 0;
            if (GEN) {
                xprev2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w[2], 0x138, 0xf, 0xf, false);
                if (lane == 0) xprev2 = cr2;
                cr2 = (uint32_t)__shfl((int)w[2], 63);
            }
            PackedCodes pc;
            packed_codes<GEN>(w, xprev, xprev2, tp, k, pc);
            mask_tail(pc, nv);
            const uint32_t lane_bits = lane_tile_bits(pc);
            const uint32_t incl = wave_incl_scan_dpp(lane_bits);
            const uint32_t tile_bits = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            const uint64_t wfirst = P >> 5;  // first staged word
            emit_tile<false>(pc, buf_bits + (uint32_t)(P & 31u) + incl - lane_bits);
            P += tile_bits;
            wave_sync();
            const uint32_t nfull = (uint32_t)((P >> 5) - wfirst);
            for (uint32_t i = lane; i < nfull; i += 64) {
                const uint64_t idx = wfirst + i;
                if (idx >= skip && idx < limit && wbase + idx < cap_words) outp[wbase + idx] = buf[i];
                buf[i] = 0;
            }
            wave_sync();
            if (nfull && lane == 0) { const uint32_t cwd = buf[nfull]; buf[nfull] = 0; buf[0] = cwd; }
            wave_sync();
        }
        const uint64_t idx = P >> 5;
        if ((P & 31u) && lane == 0 && idx >= skip && idx < limit && wbase + idx < cap_words) outp[wbase + idx] = buf[0];
    };

    if (runs) {
        const uint32_t words = s_size[wv];
        if (!PACKED && (uint32_t)lane < nspans) wave_words[q.wave_base + w0 + (uint32_t)lane] = s_n[wv][lane];
        if (pos + words > out_cap) return;  // the last workgroup raises kErrCapacity
        if (fits) {
            for (uint32_t i = lane; i < words; i += 64) out[pos + i] = buf[i];
            return;
        }
        uint64_t at = pos;
        const uint32_t n_wf = PACKED ? run_wfs : nspans;
        for (uint32_t i = 0; i < n_wf; ++i) {
            const uint32_t len_i = (w0 + i + 1u == q.W) ? q.n_samples - (w0 + i) * q.L : q.L;
            const uint64_t bits = stream_waveform(wf_off + (uint64_t)i * q.L, len_i, out + at + 1u);
            const uint32_t n = (uint32_t)((bits + 31u) >> 5);
            if (lane == 0) out[at] = n;  // src/deltaRice.c:379
            wave_sync();
            at += 1u + n;
        }
        return;
    }

    if (SUPER) {
        uint32_t *__restrict__ outp = out + pos + 1u;
        const uint64_t cap_words = out_cap > pos + 1u ? out_cap - pos - 1u : 0ull;  // payload words that still fit
        if (last_part && wv == 0u && lane == 0) {
            wave_words[q.wave_base + w0] = (uint32_t)s_wf_words;
            if (pos < out_cap) out[pos] = (uint32_t)s_wf_words;  // src/deltaRice.c:379
        }
        uint32_t before = 0;
#pragma unroll
        for (uint32_t i = 0; i < kPcWaves; ++i) before += i < wv ? s_size[i] : 0u;
        if (!part_fit) {
            // every wavefront its own segment once more (round 3; one wavefront used to take the whole part: 6 x slower)
            const uint32_t s_begin = (part * kPcWaves + wv) * sh.seg_len;
            if (s_begin < wf_len) {
                const uint32_t s_end = wf_len - s_begin < sh.seg_len ? wf_len : s_begin + sh.seg_len;
                stream_samples(s_begin, s_end, s_excl_bits + before, s_size[wv], outp, cap_words);
            }
            return;
        }
        // my words of the waveform's stream: those whose FIRST bit lies in my segment.  Bits behind my segment come from the
        // next wavefront's buffer, or (last wavefront of the part) from the second span in my own
        const uint32_t my_bits = s_size[wv];
        const uint64_t B = s_excl_bits + before, E = B + my_bits;
        const uint64_t w_lo = (B + 31u) >> 5, w_hi = (E + 31u) >> 5;
        const uint32_t *nbuf = buf_all[(wv + 1u) & (kPcWaves - 1u)] + 4;
        const bool has_next = wv + 1u < kPcWaves;
        for (uint64_t w = w_lo + (uint32_t)lane; w < w_hi; w += 64u) {
            const uint32_t o = (uint32_t)(32u * w - B), a = o >> 5, r = o & 31u;
            uint32_t v = r ? __builtin_amdgcn_alignbit(buf[a], buf[a + 1u], 32u - r) : buf[a];
            const uint32_t nb = (uint32_t)(E - 32u * w);  // my bits in this word from its top (>= 1)
            if (nb < 32u && has_next) v |= nbuf[0] >> nb;
            if (w < cap_words) outp[w] = v;
        }
        return;
    }

    const uint32_t n = (grp_bits + 31u) >> 5;  // the waveform's payload words
    if (sg == 0u && lane == 0) wave_words[q.wave_base + w0] = n;
    if (pos + 1u + n > out_cap) return;
    if (sg == 0u && lane == 0) out[pos] = n;
    uint32_t *__restrict__ outp = out + pos + 1u;
    if (!grp_fit) {  // every wavefront of the group its own segment once more
        const uint32_t s_begin = sg * sh.seg_len;
        if (s_begin < wf_len) {
            const uint32_t s_end = wf_len - s_begin < sh.seg_len ? wf_len : s_begin + sh.seg_len;
            stream_samples(s_begin, s_end, bits_before, s_size[wv], outp, out_cap - pos - 1u);
        }
        return;
    }
    // my words of the waveform's stream: those whose FIRST bit lies in my segment
    const uint32_t my_bits = s_size[wv];
    const uint32_t B = bits_before, E = bits_before + my_bits;
    const uint32_t w_lo = (B + 31u) >> 5, w_hi = (E + 31u) >> 5;
    const uint32_t *nbuf = buf_all[(wv + 1u) & (kPcWaves - 1u)] + 4;  // segment sg + 1 (read only where it exists)
    const bool has_next = sg + 1u < S;
    for (uint32_t w = w_lo + (uint32_t)lane; w < w_hi; w += 64u) {
        const uint32_t o = 32u * w - B, a = o >> 5, r = o & 31u;
        uint32_t v = r ? __builtin_amdgcn_alignbit(buf[a], buf[a + 1u], 32u - r) : buf[a];
        const uint32_t nb = E - 32u * w;  // my bits in this word from its top (>= 1)
        if (nb < 32u && has_next) v |= nbuf[0] >> nb;
        outp[w] = v;
    }
}

// The batches this encoder takes: delta filter or a forward filter of up to four taps, every WaveformLength from kPcMinLen, and a shape
// k_encode_fused is bad at somewhere (a chunk of short or of long waveforms).  Measured against the segment encoder at 100 /
// 5 / 1 chunks of 14 M samples (GB/s of int16): L = 512 2085 / 1374 / 669 against 973 / 592 / 220, L = 2048 2139 / 1494 / 664
// against 1633 / 1125 / 482, L = 16 384 2299 / 1364 / 702 against 1791 / 1265 / 504, L = 65 536 2319 / 1400 / 669 against
// 1629 / 1237 / 511 (profiles/r02_notes.md): no lower bound on the batch size.  A batch whose WaveformLengths are ALL above
// kPcMaxLen (the reference's default, one waveform per chunk) takes the SUPER form; one that mixes the two kinds stays with
// the segment encoder.
// debug_flags: 4096 never this encoder, 8192 always the segment encoder, 32768 this encoder also where WaveformLength is in
// k_encode_fused's own range (one waveform per wavefront; the tests compare the two that way).
static bool pieces_packed(const Geom &G) { return G.uniform ? piece_packable(G.u_wave_len) : G.pc_packed != 0; }
static bool pieces_super(const Geom &G) { return G.uniform ? G.u_wave_len > pc_max_len(G.k) : G.pc_super != 0; }

bool pieces_batch(const Geom &G) {
    if ((G.n_taps && !G.enc_fast) || (G.dbg & (8192u | 4096u))) return false;  // delta, or a forward filter of up to four taps
    const bool force = (G.dbg & 32768u) != 0;
    if (G.uniform) {
        const uint32_t L = G.u_wave_len;
        const bool packed = piece_packable(L);
        if (L < kPcMinLen && !packed) return false;
        if ((uint64_t)G.n_chunks * G.u_n_samples < (uint64_t)kTile) return false;
        const PieceShape sh = piece_shape(L, G.u_n_waves, G.k, packed);
        if ((uint64_t)G.u_n_waves * sh.parts > 0x7fffffffull || (uint64_t)sh.wgs * G.n_chunks > 0x7fffffffull) return false;
        if (!force && fused_wide(G)) return false;  // k_encode_fused with a larger buffer per waveform
        return force || packed || sh.run > 1u || sh.segs > 1u;
    }
    return G.pc_wg_base != nullptr;  // decided when the plan was made
}

uint64_t pieces_workgroups(const Geom &G, const ChunkDesc *host_chunks) {
    const bool packed = pieces_packed(G);
    if (G.uniform) return (uint64_t)piece_shape(G.u_wave_len, G.u_n_waves, G.k, packed).wgs * G.n_chunks;
    uint64_t t = 0;
    for (uint64_t c = 0; c < G.n_chunks; ++c) t += piece_shape(host_chunks[c].wave_len, host_chunks[c].n_waves, G.k, packed).wgs;
    return t;
}

// look-back state: one entry per workgroup (SUPER: per waveform) | SUPER: one per workgroup for the parts | ticket
uint64_t pieces_scan_words(const Geom &G, uint64_t total_wgs) {
    return (pieces_super(G) ? G.total_waves + total_wgs : total_wgs) + 2u;
}

hipError_t launch_encode_pieces(const Geom &G, const int16_t *d_in, uint64_t in_samples, uint32_t *d_out, uint64_t out_cap,
                                uint64_t *d_chunk_word_off, uint32_t *d_wave_words, uint64_t *d_scan, uint64_t total_wgs,
                                DevStatus *d_status, hipEvent_t *ev, hipStream_t s) {
    if (G.total_waves == 0 || total_wgs == 0) return hipSuccess;
    if (ev) (void)hipEventRecord(ev[0], s);
    // zeroed on the stream before every launch (an entry is its own ready flag)
    const uint64_t words = pieces_scan_words(G, total_wgs);
    hipError_t e = hipMemsetAsync(d_scan, 0, words * sizeof(uint64_t), s);
    if (e != hipSuccess) return e;
    if (ev) { (void)hipEventRecord(ev[1], s); (void)hipEventRecord(ev[2], s); }
    uint32_t *ticket = reinterpret_cast<uint32_t *>(d_scan + words - 2u);
    auto go = [&](auto super_tag, auto gen_tag, auto packed_tag) {
        constexpr bool SUPER = decltype(super_tag)::value, GEN = decltype(gen_tag)::value, PACKED = decltype(packed_tag)::value;
        k_encode_pieces<SUPER, GEN, PACKED><<<(unsigned)total_wgs, 64 * kPcWaves, 0, s>>>(
            G, d_in, in_samples, d_out, out_cap, d_chunk_word_off, d_wave_words, d_scan, SUPER ? d_scan + G.total_waves : nullptr, ticket,
            (uint32_t)total_wgs, d_status);
    };
    const bool sup = pieces_super(G), gen = G.n_taps != 0, packed = pieces_packed(G);
    const std::true_type T;
    const std::false_type F;
    if (sup) { if (gen) go(T, T, F); else go(T, F, F); }
    else if (packed) { if (gen) go(F, T, T); else go(F, F, T); }
    else { if (gen) go(F, T, F); else go(F, F, F); }
    if (ev) (void)hipEventRecord(ev[3], s);
    return hipGetLastError();
}

}  // namespace drx
