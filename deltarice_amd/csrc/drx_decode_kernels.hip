// drx_decode_kernels.hip -- gfx950 (MI355X, CDNA4) DECODE kernels of the Delta-Rice codec, the decode launch and its dispatch
// (round 3 split drx_kernels.hip by role: drx_encode_kernels.hip, drx_walk.h, this file; few long waveforms: drx_blocks.hip,
// general filters behind it: drx_iir.hip).
//
// Format contract (bit-exact with /root/reference/src/deltaRice.c):
//   chunk   := u32 N | { u32 n_i | u32 payload_i[n_i] }            (:415,379,427-433)
//   payload := MSB-first concatenation of one code per sample      (:229-241)
//   code(z) := (z>>k) zeros, '1', k bits      if (z>>k) < 8        (:215-222)
//              8 zeros, '1', 16 bits of z     otherwise            (:223-228)
//   z = zigzag(d), d_0 = x_0, d_j = x_j - x_{j-1} mod 2^16         (:51-63,207-211)
//
//
// The Rice parse is serial inside a waveform, so the unit of parallelism is the waveform: one lane per waveform, 64
// waveforms per wavefront; the 64 compressed streams are staged into per-lane LDS rings and the decoded samples are
// transposed through LDS so that HBM sees 16-byte stores of whole 128-byte lines per waveform.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "drx_internal.h"
#include "drx_device.h"
#include "drx_walk.h"

namespace drx {

// Side-band decode (drx_decode_with_wave_words): the caller hands over the n_i table an encode left behind (SURVEY section 7:
// "reuse its offset table as a side-band"), so no header chain is walked.  Per chunk: header positions by a prefix sum over
// 1 + n_i, each checked against the stream itself (the word at that position must BE n_i, n_i within the bounds of its
// waveform, the chain must end exactly at the chunk's end, the chunk header must be the sample count) -- a table that does
// not belong to the stream is DRX_ERR_CORRUPT, never a wild read.
__global__ __launch_bounds__(256) void k_sideband_tables(Geom G, const uint32_t *__restrict__ in, uint64_t in_words,
                                                         const uint64_t *__restrict__ chunk_word_off,
                                                         const uint32_t *__restrict__ n_in, uint64_t *__restrict__ wave_off,
                                                         uint32_t *__restrict__ wave_words, DevStatus *st) {
    __shared__ uint64_t wsum[4];
    const uint64_t c = blockIdx.x;
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    uint64_t base;
    uint32_t W, L, N;
    if (G.uniform) { base = c * G.u_n_waves; W = G.u_n_waves; L = G.u_wave_len; N = G.u_n_samples; }
    else { const ChunkDesc d = G.chunks[c]; base = d.wave_base; W = d.n_waves; L = d.wave_len; N = d.n_samples; }
    const uint64_t off0 = chunk_word_off[c], off1 = chunk_word_off[c + 1];
    bool bad = off1 > in_words || off0 >= off1;  // (the same in every lane here; block-wide after every round below)
    uint64_t run = 1;  // the chunk header word
    for (uint32_t i0 = 0; i0 < W; i0 += 256) {
        const uint32_t i = i0 + threadIdx.x;
        const uint32_t n = (i < W) ? n_in[base + i] : 0u;
        const uint32_t len = (i < W) ? ((i + 1u == W) ? N - i * L : L) : 0u;
        // An entry outside the bounds of its waveform never enters the prefix sum (64-bit: 256 entries of up to 25 bits per
        // sample of a 2^31-sample waveform do not fit 32), so no later position can wrap below off0 or past 2^64; and the
        // whole chunk is rejected before any lane looks at the stream.
        const bool n_ok = i >= W || (n <= max_payload_words(len) && n >= min_payload_words(len, G.k));
        const uint64_t v = (i < W && n_ok) ? (uint64_t)n + 1u : 0u;
        uint64_t inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t t = __shfl_up(inc, d);
            if (lane >= d) inc += t;
        }
        if (lane == 63) wsum[wv] = inc;
        bad = __syncthreads_or(bad || !n_ok) != 0;
        uint64_t before = 0, all = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { before += (w < wv) ? wsum[w] : 0u; all += wsum[w]; }
        bool mine_bad = bad;
        if (i < W) {
            const uint64_t at = off0 + run + before + inc - v;
            if (!mine_bad && (at <= off0 || at >= off1 || at + 1u + n > off1)) mine_bad = true;
            if (!mine_bad && in[at] != n) mine_bad = true;
            // a rejected entry is left pointing at the chunk's own header with no payload words: the decode launch behind
            // this one runs whatever the status says, and reads nothing through such an entry
            wave_off[base + i] = mine_bad ? off0 : at;
            wave_words[base + i] = mine_bad ? 0u : n;
        }
        run += all;
        bad = __syncthreads_or(mine_bad) != 0;  // (also keeps wsum until every lane has read it)
    }
    if (threadIdx.x == 0 && !bad && (off0 + run != off1 || in[off0] != N)) bad = true;
    if (bad && threadIdx.x == 0) atomicOr(&st->err, kErrCorrupt);
}

// Straightforward lane-per-waveform decoder: global loads and 2-byte stores.
// Kept as the simple cross-check of the staged kernel below (decode_impl = 0).
__global__ __launch_bounds__(64) void k_decode_simple(Geom G, const uint32_t *__restrict__ in,
                                                      const uint64_t *__restrict__ wave_off,
                                                      const uint32_t *__restrict__ wave_words, DevStatus *st,
                                                      int16_t *__restrict__ out, const uint32_t *__restrict__ only) {
    const uint64_t g = (uint64_t)blockIdx.x * 64u + threadIdx.x;
    if (g >= G.total_waves) return;
    if (only && !only[g]) return;  // (behind the block decoder with a general filter: the waveforms it flagged)
    const WaveRef r = locate(G, g);
    const uint32_t *s = in + wave_off[g] + 1;
    const uint32_t n = wave_words[g];
    int16_t *y = out + r.sample_off;
    const uint32_t k = G.k;
    uint64_t win = 0;
    uint32_t have = 0, wi = 0;
    int32_t acc = 0;
    int16_t hist[64];  // general prediction filters only (lives in scratch; the delta path never touches it)
    for (uint32_t i = 0; i < r.len; ++i) {
        if (have <= 32u) {
            const uint32_t w = wi < n ? s[wi] : 0u;
            ++wi;
            win |= (uint64_t)w << (32u - have);
            have += 32u;
        }
        uint32_t q = (uint32_t)__clzll((long long)win);
        q = q > 8u ? 8u : q;
        const uint32_t pl = (q == 8u) ? 16u : k;
        const uint64_t t = win << (q + 1u);
        const uint32_t rem = pl ? (uint32_t)(t >> (64u - pl)) : 0u;
        const uint32_t z = (q == 8u) ? rem : ((q << k) + rem);
        const int32_t d = (int32_t)(z >> 1) ^ -(int32_t)(z & 1u);  // un-zig-zag (:172-177)
        if (G.n_taps == 0) {
            acc += d;  // running sum (:80-89)
        } else {
            // general inverse (:92-101): y[i] = (int16)((int16)(d[i] - sum_{j>=1} taps[j] y[i-j]) / taps[0]);
            // the last 64 outputs live in a lane-private circular history (taps <= DRX_MAX_TAPS = 64)
            uint32_t a = (uint32_t)(int32_t)(int16_t)d;
            for (uint32_t j = 1; j < G.n_taps && j <= i; ++j) a -= (uint32_t)((int32_t)hist[(i - j) & 63u] * G.taps[j]);
            acc = (int32_t)(int16_t)(uint16_t)a / G.taps[0];
            hist[i & 63u] = (int16_t)acc;
        }
        y[i] = (int16_t)acc;
        const uint32_t used = q + 1u + pl;
        win <<= used;
        have -= used;
    }
    // a valid waveform's codes end in its last payload word: n_i = ceil(bits / 32) (src/deltaRice.c:237-241)
    const uint64_t bits = 32ull * wi - have;
    if (r.len && ((bits + 31u) >> 5) != n) atomicOr(&st->err, kErrCorrupt);
}

// Decoder for FEW, LONG waveforms (WaveformLength = -1, the reference's default, makes every chunk one
// waveform of millions of samples): one WORKGROUP of 8 wavefronts per waveform, its 512 lanes parse 512
// consecutive 512-bit segments of the stream at once.  A lane does not know where the first code of its segment
// starts; it assumes the segment start, parses to the end of its segment and reports where its last
// code ended.  Rice codes re-synchronise within a few codes, so that end is almost always right even
// when the start was not.  Each lane then restarts from its predecessor's end until no start changes
// (lane 0's start is known, so after at most 512 rounds every lane is exact; typically one restart),
// the sample counts and delta sums of the segments are prefix-summed over the workgroup, and a last parse
// writes the samples.  Three parses of every bit instead of one, 512 at a time.  Delta filter only (the
// prefix sum over segment sums is what makes the segments independent).
constexpr int kLongSeg = 16;  // words per lane and block
constexpr int kLongOv = 2;    // words of the next segment kept below a lane's column (a code has <= 25 bits)
constexpr uint32_t kLongSegBits = kLongSeg * 32u;
constexpr int kLongWaves = 8;                 // wavefronts per workgroup
constexpr int kLongThreads = 64 * kLongWaves;  // segments parsed at once
constexpr uint32_t kLongGuessBits = 160;  // bits in front of a segment's end from which the first guess is parsed

// One workgroup per waveform walks its blocks in order (fail != nullptr: only the waveforms it flags).  This is the
// FALLBACK of the block-parallel decoder (drx_blocks.hip), which gives every block a workgroup of its own and is 2-6 x
// faster where a parse falls into step within a few codes; waveforms where that fails (a slope-1 ramp: all codes the same
// length) are flagged and come here, where a block starts exactly where its predecessor ended.  (The workgroup-per-block
// form of THIS kernel, with a tail parse predicting block starts, was round 1's path for a handful of long waveforms; the
// new decoder replaced it: 25 x 14 M samples 1.55 -> 0.83 ms, profiles/r02_notes.md.)
__global__ __launch_bounds__(kLongThreads) void k_decode_long(Geom G, const uint32_t *__restrict__ in,
                                                              const uint64_t *__restrict__ wave_off,
                                                              const uint32_t *__restrict__ wave_words, DevStatus *st,
                                                              int16_t *__restrict__ out, const uint32_t *__restrict__ fail,
                                                              const uint32_t *__restrict__ suspect, uint32_t verdict_only) {
    constexpr uint32_t NT = kLongThreads;
    // [kLongRows - 2 - word of the segment][thread] (a lane's bank is its lane number whatever row it reads), rows
    // in REVERSE word order plus one unused row on top: with the bit position kept negated, Q = -pos, the row
    // pair (Q >> 5, Q >> 5 + 1) is (word w + 1, word w) inside a word and (word w, word w - 1) on a word
    // boundary, and v_alignbit_b32(hi, lo, Q) is the 32-bit window in both cases (as in k_decode_lanes)
    // (+ 2 rows below: a lane that has left its segment keeps reading at its last position, up to 49 bits past it)
    constexpr uint32_t kLongRows = kLongSeg + kLongOv + 3;
    __shared__ uint32_t col[kLongRows * NT];
    __shared__ uint32_t s_end[NT];
    __shared__ uint32_t s_tot[2][kLongWaves];
    const uint32_t tid = threadIdx.x;
    const int wv = (int)(tid >> 6), lane = (int)(tid & 63u);
    const uint64_t g = blockIdx.x;
    if (g >= G.total_waves) return;
    if (fail && !fail[g]) {
        // not flagged: the block-parallel decoder's output stands, and so does its verdict on the stream
        if (suspect && suspect[g] && tid == 0) atomicOr(&st->err, kErrCorrupt);
        return;
    }
    if (verdict_only) return;  // (general filters: flagged waveforms are decoded again by k_decode_simple, which judges them too)
    const WaveRef r = locate(G, g);
    const uint32_t *src = in + wave_off[g] + 1;
    const uint32_t n = wave_words[g];
    int16_t *y = out + r.sample_off;
    const uint32_t len = r.len, k = G.k;
    uint32_t blk_word = 0;  // first word of the block
    uint32_t carry_in = 0;  // bit of thread 0's segment at which the next code starts
    uint32_t done = 0;      // samples written
    uint32_t acc_base = 0;  // running sum before the block (mod 2^16)

    // parses this thread's segment from bit `start`; a code is taken when it STARTS inside the segment and inside
    // the stream.  emit: add to the running sum `acc` and store sample number idx, idx + 1, ...
    auto parse = [&](bool enable, uint32_t start, uint32_t avail_bits, auto emit_tag, uint32_t idx, uint32_t acc,
                     uint32_t &end, uint32_t &cnt, uint32_t &sum) __attribute__((always_inline)) {
        constexpr bool EMIT = decltype(emit_tag)::value;
        uint32_t c = 0, sacc = EMIT ? acc : 0u;
        uint32_t Q = 0u - start;  // minus the bit position
        const uint32_t lim = avail_bits < kLongSegBits ? avail_bits : kLongSegBits;
        const int32_t nlim = enable ? -(int32_t)lim : 1;  // a code is taken while -Q < lim, i.e. Q > -lim
        // EMIT: two samples per store where they fill an aligned dword (2-byte stores run into the L2's request rate)
        const uint32_t par4 = (uint32_t)(((uintptr_t)y >> 1) & 1u);  // sample 0's half of its aligned dword
        uint32_t held = 0, held_i = 0;  // the sample waiting for its partner
        bool holding = false;
        // LDS byte address of this thread's row of word 0
        const uint32_t row0 = lds_addr(col) + ((kLongRows - 2u) * NT + tid) * 4u;
        while (__any((int32_t)Q > nlim)) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {  // one vote per four codes
                const bool act = (int32_t)Q > nlim;
                typedef const uint32_t __attribute__((address_space(3))) lds_cu32;
                const lds_cu32 *wp = (const lds_cu32 *)(uintptr_t)(row0 + (uint32_t)(((int32_t)Q >> 5) * (int32_t)(NT * 4u)));
                const uint32_t lo = wp[0], hi = wp[NT];
                const uint32_t win = __builtin_amdgcn_alignbit(hi, lo, Q);
                const uint32_t q = ffbh(win);  // all-zero window (padding): the hardware's -1 is as good as any
                const uint32_t kk = (win < (1u << 24)) ? 16u : k;
                const uint32_t nu = ~(q + kk);  // minus the code length
                const uint32_t z = (q << kk) + __builtin_amdgcn_ubfe(win, nu, kk);
                const uint32_t d = (z >> 1) ^ (0u - (z & 1u));
                const uint32_t s2 = sacc + d;
                if (EMIT) {
                    if (act && idx + c < len && !(kAblate && (G.dbg & 16384u))) {  // (16384: ablation, no stores)
                        const uint32_t i = idx + c;
                        if (((i + par4) & 1u) == 0u) {  // low half of an aligned dword: wait for the next sample
                            held = s2 & 0xffffu;
                            held_i = i;
                            holding = true;
                        } else if (holding) {
                            *reinterpret_cast<uint32_t *>(y + i - 1u) = held | (s2 << 16);
                            holding = false;
                        } else {
                            y[i] = (int16_t)(uint16_t)s2;  // the lane's first sample sits in a high half
                        }
                    }
                }
                sacc = act ? s2 : sacc;
                Q = act ? Q + nu : Q;
                c += act ? 1u : 0u;
            }
        }
        if (EMIT && holding) y[held_i] = (int16_t)(uint16_t)held;  // the lane's last sample had no partner
        const uint32_t pos = 0u - Q;
        end = pos;
        cnt = c;
        sum = sacc;
    };

    while (done < len && blk_word < n) {
        // the block's words, transposed into per-thread columns; the first kLongOv words of segment s + 1 repeat below column s
#pragma unroll
        for (int rr = 0; rr < kLongSeg; ++rr) {
            const uint32_t j = tid + NT * (uint32_t)rr;
            const uint32_t wvl = (blk_word + j < n) ? src[blk_word + j] : 0u;
            const uint32_t sgm = j / kLongSeg, i = j % kLongSeg;
            col[(kLongRows - 2u - i) * NT + sgm] = wvl;
            if (i < (uint32_t)kLongOv && sgm >= 1u) col[(kLongRows - 2u - ((uint32_t)kLongSeg + i)) * NT + sgm - 1u] = wvl;
        }
        if (tid < (uint32_t)kLongOv) {
            const uint32_t wi = blk_word + NT * kLongSeg + tid;
            col[(kLongRows - 2u - ((uint32_t)kLongSeg + tid)) * NT + NT - 1u] = (wi < n) ? src[wi] : 0u;
        }
        __syncthreads();
        const uint32_t seg_word = blk_word + tid * kLongSeg;
        const uint32_t avail_bits = seg_word < n ? ((n - seg_word) > (1u << 26) ? 0xffffffffu : (n - seg_word) * 32u) : 0u;

        // first guess: only where the segment's last code ends is wanted, and a parse re-synchronises within a few
        // codes, so the guess starts kLongGuessBits before the segment's end (thread 0 knows its start)
        uint32_t start = tid == 0 ? carry_in : kLongSegBits - kLongGuessBits, end, cnt, sum;
        parse(true, start, avail_bits, std::false_type{}, 0u, 0u, end, cnt, sum);
        for (uint32_t it = 0; it < NT; ++it) {
            s_end[tid] = end;
            __syncthreads();
            // where the predecessor's last code ended, as a bit of MY segment (kLongSegBits = "nothing left for me")
            const uint32_t pe = tid ? s_end[tid - 1u] : 0u;
            const uint32_t ns = tid == 0 ? carry_in : (pe >= kLongSegBits ? pe - kLongSegBits : kLongSegBits);
            const bool changed = ns != start;
            if (!__syncthreads_or(changed ? 1 : 0)) break;  // (also orders the reads of s_end before its next writes)
            start = ns;
            uint32_t e2, c2, s2;
            parse(changed, start, avail_bits, std::false_type{}, 0u, 0u, e2, c2, s2);
            if (changed) { end = e2; cnt = c2; sum = s2; }
        }
        // prefix sums over the workgroup: samples before my segment, sum of deltas before my segment
        const uint32_t incl_c = wave_incl_scan_dpp(cnt), incl_s = wave_incl_scan_dpp(sum);
        if (lane == 63) { s_tot[0][wv] = incl_c; s_tot[1][wv] = incl_s; }
        s_end[tid] = end;
        __syncthreads();
        uint32_t pre_c = 0, pre_s = 0, tot_c = 0, tot_s = 0;
#pragma unroll
        for (int i = 0; i < kLongWaves; ++i) {
            const uint32_t tc = s_tot[0][i], ts = s_tot[1][i];
            if (i < wv) { pre_c += tc; pre_s += ts; }
            tot_c += tc;
            tot_s += ts;
        }
        const uint32_t end_last = s_end[NT - 1u];
        uint32_t e3, c3, s3;
        parse(true, start, avail_bits, std::true_type{}, done + pre_c + incl_c - cnt, acc_base + pre_s + incl_s - sum, e3, c3, s3);
        done = (tot_c > len - done) ? len : done + tot_c;
        acc_base += tot_s;
        carry_in = end_last >= kLongSegBits ? end_last - kLongSegBits : 0u;
        blk_word += NT * kLongSeg;
        __syncthreads();
    }
    if (done < len && tid == 0) atomicOr(&st->err, kErrCorrupt);  // the stream ended before the waveform did
}

// Staged lane-per-waveform decoder (the production kernel; generations 1-4 are in the git
// history, what each measurement changed is in DESIGN.md).
//
// Decomposition.  The Rice parse is serial inside a waveform, so the unit of parallelism is the
// waveform: one lane per waveform, 64 waveforms per wavefront.
//   stream in   each lane owns an LDS ring of RW words of its compressed stream, refilled in
//               pieces of LW words by 16-byte loads (every byte of the stream is requested once;
//               rocprofv3: TCC_EA0_RDREQ x 128 B = the stream size).  A piece is loaded ahead of
//               need and written to the ring just before the round's stores are issued (vmcnt is
//               in order: a load issued after a store cannot be waited for without that store).
//   ring layout word-major and reversed, ring[RW - (w mod RW)][lane], plus a mirror row: a lane's
//               bank is its lane number whatever row it reads (no conflicts although the 64
//               streams drift apart), and the pair (w, w+1) is always (row+1, row).
//   per sample  the bit position is kept negated, Q = -P: row = Q[5 +: log2 RW] (v_bfe),
//               (lo, hi) = ds_read2st64_b32, win = v_alignbit(hi, lo, Q): 3 VALU + 1 LDS
//               instruction form the 32-bit window, no refill state.  Escape and ordinary codes
//               share one extraction (payload width kk = esc ? 16 : k; the 8 << 16 an escape leaves
//               above bit 15 never reaches the int16 running sum).  14.5 VALU instructions per
//               sample; the kernel is bound by VALU issue (4 cycles per wave64 instruction).
//   samples out transposed through LDS; a lane-private start delay phi makes step u of every round
//               land u*2 bytes past a T*2-byte boundary, so each round stores whole aligned
//               128-byte lines (T = 64).  tools/ubench_store.hip: 16-byte stores reach 5.5 TB/s only
//               when every contiguous run is a whole line; 64-byte aligned runs give 4.2 TB/s and
//               runs that straddle lines 2.6-3.3 TB/s whatever their length.
//   PAIR        two samples per ring access: three words = a 64-bit window always hold two codes
//               (2 x 25 bits); the second sample's window is v_alignbit(winA, winB, -len1).  One LDS
//               round trip on the dependent chain per two samples.
//   FUSED       the header-chain walk runs inside the same launch: workgroups take a ticket; the first
//               ceil(n_chunks/8) tickets walk (eight chunks per wave through scalar loads,
//               walk_chunks_scalar(), publishing one granule per waveform), every later ticket decodes
//               64 waveforms of one chunk as soon as their granules appear.  Decode tickets are dealt group-major (waveforms 0-63 of every chunk,
//               then 64-127 of every chunk, ...), the order in which the 2000-hop chains release them,
//               so the ~1.7 ms of dependent-load latency of the walk disappears behind the decode.
//               A ticket holder is by construction running, so waiting on a lower ticket's walker
//               cannot deadlock whatever the dispatch order.  (Uniform batches only.)
//   NW > 1      STAGED FLUSH (round 3).  What limits the wavefronts per CU is LDS: 16.9 KB of ring + 9.2 KB of transposition
//               buffer allow six, and at an unchanged instruction stream 6 -> 9 per CU is worth 21 % (profiles/r03_notes.md,
//               occupancy A/B).  The transposition buffer is needed only while a round's samples change hands, so NW
//               wavefronts form one workgroup and SHARE one: a wavefront keeps the round's 64 samples per lane in 32
//               VGPRs (the interior rounds are fully unrolled: static register indices), takes the workgroup's lock
//               (an LDS compare-and-swap by lane 0), dumps its registers with eight ds_write_b128, reads them back
//               transposed with eight ds_read_b128, releases the lock and stores whole lines as before.  Rows are 128
//               bytes without padding; the 16-byte block j of row r lives at block j ^ (r & 7), which makes both the
//               dump and the transposed read conflict-free.  The rare edge rounds (masked stores, per-sample staging)
//               hold the lock for the whole round and use the buffer as the private one was used.  NW = 4: two
//               workgroups = 8 wavefronts per CU; NW = 9: one workgroup = 9 per CU (160 260 of 163 840 bytes).
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

template <int RW, int LW, int T, int GS, bool FUSED, bool PAIR = false, bool GEN = false, int NW = 1>
__global__ __launch_bounds__(64 * NW, NW == 1 ? 1 : (NW == 4 ? 2 : 3)) void k_decode_lanes(Geom G, const uint32_t *__restrict__ in, uint64_t in_words,
                                                     const uint64_t *__restrict__ chunk_word_off,
                                                     uint64_t *__restrict__ wave_off,
                                                     uint32_t *__restrict__ wave_words,
                                                     uint64_t *__restrict__ granules, uint32_t *__restrict__ ticket,
                                                     DevStatus *st, int16_t *__restrict__ out) {
    static_assert(RW >= 2 * LW && (RW == 32 || RW == 64 || RW == 128) && (LW == 8 || LW == 16 || LW == 32), "ring");
    constexpr int LOG_RW = RW == 32 ? 5 : (RW == 64 ? 6 : 7);
#ifndef DRX_DEC_OPAD
#define DRX_DEC_OPAD 4
#endif
    constexpr bool STG = NW > 1;
    static_assert(!STG || (PAIR && T == 64 && GS == 16), "the staged flush is built for two samples per access and 64-sample rounds");
    constexpr int OSW = T / 2 + (STG ? 0 : DRX_DEC_OPAD);  // output row stride in words (16- or 8-byte aligned rows)
    static_assert(OSW % 2 == 0, "rows are read in 8- or 16-byte pieces");
    constexpr int PPS = T / 8;      // 16-byte pieces per stream per round
    constexpr int SPI = 64 / PPS;   // streams per write-out iteration
    constexpr int NV = LW / 4;      // 16-byte loads per piece
    constexpr uint32_t WMASK = (1u << 27) - 1u;
    constexpr uint32_t NEED_AT = GS + 2;                                   // must refill below this many words
    static_assert(T % GS == 0 && RW - LW >= GS + 2, "round length / ring slack");
    // row r: word w with RW - (w mod RW) == r; row 0 mirrors row RW and row -1 mirrors row RW - 1 (PAIR reads
    // three consecutive words: rows r + 1, r, r - 1)
    __shared__ uint32_t ring_mem[NW][(RW + 2) * 64];
    const int wv = STG ? (int)(threadIdx.x >> 6) : 0;
    uint32_t (&ring_all)[(RW + 2) * 64] = ring_mem[wv];
    uint32_t *const ring = ring_all + 64;
    __shared__ uint32_t fl_lock;  // STG: who holds obuf (0: nobody)
#ifdef DRX_DEC_NOOBUF  // (ablation builds: no transposition buffer, a lane's sample stores go to four words, no write-out)
    constexpr int OBW = 256;
    constexpr bool kNoObuf = true;
#else
    constexpr int OBW = 64 * OSW;
    constexpr bool kNoObuf = false;
#endif
    __shared__ __attribute__((aligned(16))) uint32_t obuf[OBW];  // doubles as the start-up tables
    uint64_t *tab_off = reinterpret_cast<uint64_t *>(obuf);  // [64] sample offset of step 0 of round 0
    uint32_t *tab_lo = obuf + 128, *tab_hi = obuf + 192;      // [64] each
    static_assert(64 * OSW >= 256, "tables fit in obuf");

    const int lane = lane_id();
    const uint32_t k = G.k;
    if constexpr (STG) {  // (before any wavefront can leave: a walker returns early)
        if (threadIdx.x == 0) fl_lock = 0u;
        __syncthreads();
    }
    // the workgroup's lock on obuf: taken by lane 0, held by the wavefront.  LDS operations of a wavefront complete in
    // order, so the holder's reads are done once its s_waitcnt has passed; the holder never waits for another wavefront.
    auto flush_lock = [&]() __attribute__((always_inline)) {
        if constexpr (STG) {
            if (kAblate && (G.dbg & 8u)) return;  // (ablation: no lock -- the samples of neighbouring wavefronts mix)
            for (;;) {
                uint32_t got = 1u;
                if (lane == 0) {
                    uint32_t expect = 0u;
                    got = __hip_atomic_compare_exchange_strong(&fl_lock, &expect, 1u, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED,
                                                               __HIP_MEMORY_SCOPE_WORKGROUP) ? 0u : 1u;
                }
                if ((uint32_t)__builtin_amdgcn_readfirstlane((int)got) == 0u) break;
                __builtin_amdgcn_s_sleep(1);
            }
            wave_sync();
        }
    };
    auto flush_unlock = [&]() __attribute__((always_inline)) {
        if constexpr (STG) {
            if (kAblate && (G.dbg & 8u)) return;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // every read of the buffer has returned
            if (lane == 0) __hip_atomic_store(&fl_lock, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };
    // Stores go through pointers with an explicit global address space: once `out` has travelled through
    // nested by-reference lambda captures the compiler no longer infers it and emits flat_store, which
    // also ticks lgkmcnt and serialises against the LDS traffic of the write-out (measured: 2x slower).
    typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
    typedef u32x4v __attribute__((address_space(1))) g_uint4;
    typedef int16_t __attribute__((address_space(1))) g_i16;
    g_i16 *const outg = (g_i16 *)out;
    uint64_t g;
    bool active;
    uint32_t len = 0, n = 0;
    uint64_t S = 1, ooff = 0;
    const uint32_t wave_idx = STG ? blockIdx.x * (uint32_t)NW + (uint32_t)wv : blockIdx.x;  // (outside the ticketed launch)
    if (FUSED) {
        uint32_t tk = 0;
        if (lane == 0) tk = atomicAdd(ticket, 1u);
        tk = (uint32_t)__builtin_amdgcn_readfirstlane((int)tk);
        // walker tickets first.  Chunks of short waveforms (tens of thousands of hops) are streamed
        // through LDS by a whole wave each (walk_chunk_block, in this wave's ring/transposition LDS),
        // the others are chased through scalar loads, kWalkChains chunks per wave.
        constexpr bool kBlockWalkFits = !STG && sizeof(ring_all) >= kWalkBlockWords * 4u && sizeof(obuf) >= kWalkHopCap * 8u;
        const bool u_short = G.uniform && G.u_wave_len <= kWalkShortLen;
        const uint32_t n_blockwalk = G.uniform ? (u_short ? (uint32_t)G.n_chunks : 0u) : G.n_short;
        const uint64_t n_chain = G.uniform ? (u_short ? 0ull : G.n_chunks) : (uint64_t)G.n_long;
        const uint32_t n_walk = n_blockwalk + (uint32_t)((n_chain + (uint32_t)kWalkChains - 1u) / (uint32_t)kWalkChains);
        if (tk < n_walk) {  // walker role
            __builtin_amdgcn_s_setprio(3);  // the chain is the critical path of the whole launch (A/B: -2.5 %)
            if (tk < n_blockwalk) {
                if constexpr (kBlockWalkFits) {
                    const uint64_t c = G.uniform ? (uint64_t)tk : (uint64_t)G.walk_short[tk];
                    walk_chunk_block(G, c, in, in_words, chunk_word_off, wave_off, wave_words, granules, st, ring_all,
                                     reinterpret_cast<uint2 *>(obuf));
                }
            } else {
                walk_chunks_scalar(G, (uint64_t)(tk - n_blockwalk) * kWalkChains, G.uniform ? nullptr : G.walk_long, n_chain,
                                   in, in_words, chunk_word_off, wave_off, wave_words, granules, st);
            }
            return;
        }
        // decode tickets, group-major: waveforms 0-63 of every chunk, then 64-127 of every chunk, ...
        const uint64_t t2 = tk - n_walk;
        uint64_t grp = t2 / G.n_chunks, c = t2 - grp * G.n_chunks;
        uint32_t idx = (uint32_t)grp * 64u + lane;  // waveform index inside chunk c
        uint64_t gr = 0;
        if (G.uniform) {
            // The chunks' LAST groups are partial (2000 waveforms: 31 full groups + 16 waveforms): several chunks' leftovers
            // share a wavefront -- 4 x 16 lanes for the headline batch, 15 625 wavefronts instead of 16 000, the last 500 of
            // them a quarter full (1920 against 2000 waveforms per chunk had measured 4 %).  Everything below is per lane.
            const uint32_t full = G.u_n_waves >> 6, rem = G.u_n_waves & 63u;
            const uint64_t t_full = (uint64_t)full * G.n_chunks;
            if (t2 < t_full) {
                active = true;
            } else {
                if (rem == 0u) return;  // (STG: a ticket beyond the last group)
                const uint32_t per = 64u / rem, sub = (uint32_t)lane / rem;
                const uint64_t c0 = (t2 - t_full) * per;
                if (c0 >= G.n_chunks) return;
                c = c0 + sub;
                idx = full * 64u + ((uint32_t)lane - sub * rem);
                active = sub < per && c < G.n_chunks;
                if (!active) { c = c0; idx = 0; }
            }
            g = c * G.u_n_waves + idx;
            if (active) {
                len = (idx + 1 == G.u_n_waves) ? (G.u_n_samples - idx * G.u_wave_len) : G.u_wave_len;
                ooff = c * (uint64_t)G.u_n_samples + (uint64_t)idx * G.u_wave_len;
            }
        } else {
            // ragged: the grid covers max_groups groups of every chunk; a chunk with fewer has idle tickets
            const ChunkDesc d = G.chunks[c];
            if ((uint32_t)grp * 64u >= d.n_waves) return;
            active = idx < d.n_waves;
            g = d.wave_base + idx;
            if (active) {
                len = (idx + 1 == d.n_waves) ? (d.n_samples - idx * d.wave_len) : d.wave_len;
                ooff = d.sample_off + (uint64_t)idx * d.wave_len;
            }
        }
        uint32_t spins = 0;
        for (;;) {  // wait for this wave's granules; the walker that writes them holds a lower ticket
            if (active && !(gr & kGranValid))
                gr = __hip_atomic_load(granules + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!__any(active && !(gr & kGranValid))) break;
            __builtin_amdgcn_s_sleep(8);
            if (++spins > (1u << 24)) {  // cannot happen; never hang the GPU
                if (lane == 0) atomicOr(&st->err, kErrInternal);
                gr |= kGranValid;
                break;
            }
        }
        if (active) {
            n = (uint32_t)(gr >> 32) & 0x7fffffffu;
            S = chunk_word_off[c] + (uint32_t)gr + 1u;
        }
    } else if (!G.uniform && G.rag_order) {
        // ragged batch: wavefronts in order of decreasing WaveformLength (longest processing time first).  A lane takes
        // ~60 ns per sample whatever else runs, so a wavefront of 16 384-sample waveforms that starts last adds its
        // whole 1 ms to the launch (config 5: 1.9 -> 1.2 ms)
        if (wave_idx >= G.rag_groups) return;  // (STG: the last workgroup may have wavefronts to spare)
        const uint2 e = G.rag_order[wave_idx];  // {chunk, group of 64 waveforms inside it}
        const ChunkDesc d = G.chunks[e.x];
        const uint32_t idx = e.y * 64u + (uint32_t)lane;
        active = idx < d.n_waves;
        g = d.wave_base + idx;
        if (active) {
            len = (idx + 1 == d.n_waves) ? (d.n_samples - idx * d.wave_len) : d.wave_len;
            ooff = d.sample_off + (uint64_t)idx * d.wave_len;
            S = wave_off[g] + 1u;
            n = wave_words[g];
        }
    } else {
        g = (uint64_t)wave_idx * 64u + lane;
        active = g < G.total_waves;
        if (active) {
            const WaveRef r = locate(G, g);
            len = r.len;
            ooff = r.sample_off;
            S = wave_off[g] + 1u;
            n = wave_words[g];
        }
    }
    // start delay: step u of every round sits u*2 bytes past a T*2-byte boundary
    const uint32_t phi = active ? (uint32_t)((((uintptr_t)out >> 1) + ooff) & (uint64_t)(T - 1)) : 0u;
    const uint32_t steps = wave_max_u32(len + phi);
    const uint32_t lo_max = wave_max_u32(phi);
    const uint32_t hi_min = ~wave_max_u32(~(phi + len));
    uint64_t wo_off[PPS];  // write-out constants: piece p of stream st_i, i = 0..PPS-1
    uint32_t wo_lo[PPS], wo_hi[PPS];
    // STG: lines are addressed as a wave-uniform base (the line of the wavefront's first waveform: an SGPR pair) plus a
    // 32-bit offset per lane -- half the registers, and no 64-bit vector adds in front of the stores (the launcher sends
    // batches whose 64 waveforms could lie 2^31 samples apart to the one-wavefront form)
    const uint64_t line0 = ooff - phi;
    const uint64_t line_base = STG ? (((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(line0 >> 32)) << 32) |
                                      (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)line0))
                                   : 0ull;
    uint32_t wo_rel[PPS];
    if constexpr (STG) {
#pragma unroll
        for (int i = 0; i < PPS; ++i) {
            const int st = i * SPI + lane / PPS, p = lane % PPS;
            wo_rel[i] = (uint32_t)__shfl((int)(uint32_t)(line0 - line_base), st) + 8u * (uint32_t)p;
            wo_off[i] = 0;
            wo_lo[i] = wo_hi[i] = 0u;
        }
    } else {
        tab_off[lane] = ooff - phi;
        tab_lo[lane] = phi;
        tab_hi[lane] = phi + len;
        wave_sync();
#pragma unroll
        for (int i = 0; i < PPS; ++i) {
            const int st = i * SPI + lane / PPS, p = lane % PPS;
            wo_off[i] = tab_off[st] + 8u * (uint32_t)p;
            wo_lo[i] = tab_lo[st];
            wo_hi[i] = tab_hi[st];
        }
        wave_sync();
    }

    const uint64_t A = (S & ~(uint64_t)(RW - 1)) - (uint64_t)RW;  // s0 in [RW, 2 RW)
    const uint32_t s0 = (uint32_t)(S - A);
    const uint32_t endw = s0 + n;
    uint32_t flw = s0 & ~(uint32_t)(LW - 1);
    const bool in_vec_ok = ((uintptr_t)in & 15u) == 0;
    uint32_t *myring = ring + lane;
    typedef uint16_t __attribute__((may_alias)) u16a;
#ifdef DRX_DEC_NOOBUF
    u16a *myout = reinterpret_cast<u16a *>(obuf + lane * 4);
#define DRX_OIDX(x) ((x) & 6)
#else
    u16a *myout = reinterpret_cast<u16a *>(obuf + lane * OSW);
#define DRX_OIDX(x) (x)
#endif
    // the round's whole-line stores (A/B: -DDRX_DEC_NT_STORE marks them non-temporal)
    auto store16 = [&](g_i16 *dst, const uint4 &v) __attribute__((always_inline)) {
#ifdef DRX_DEC_NT_STORE
        __builtin_nontemporal_store((u32x4v){v.x, v.y, v.z, v.w}, (g_uint4 *)dst);
#else
        *(g_uint4 *)dst = (u32x4v){v.x, v.y, v.z, v.w};
#endif
    };
    // 16 bytes of a stream's row: one ds_read_b128 where the rows are 16-byte aligned, else two ds_read_b64
    auto orow16 = [&](int st, int p) __attribute__((always_inline)) -> uint4 {
        if constexpr (OSW % 4 == 0) {
            return *reinterpret_cast<const uint4 *>(obuf + st * OSW + 4 * p);
        } else {
            const uint2 a = *reinterpret_cast<const uint2 *>(obuf + st * OSW + 4 * p);
            const uint2 b = *reinterpret_cast<const uint2 *>(obuf + st * OSW + 4 * p + 2);
            return make_uint4(a.x, a.y, b.x, b.y);
        }
    };

    auto load_piece = [&](uint4 (&v)[NV], uint32_t ahead = 0) {
        const uint64_t a = A + flw + ahead;
        if (in_vec_ok && a + (uint32_t)LW <= in_words) {
#pragma unroll
            for (int j = 0; j < NV; ++j) v[j] = *reinterpret_cast<const uint4 *>(in + a + 4 * j);
        } else {
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                v[j].x = (a + 4 * j + 0 < in_words) ? in[a + 4 * j + 0] : 0u;
                v[j].y = (a + 4 * j + 1 < in_words) ? in[a + 4 * j + 1] : 0u;
                v[j].z = (a + 4 * j + 2 < in_words) ? in[a + 4 * j + 2] : 0u;
                v[j].w = (a + 4 * j + 3 < in_words) ? in[a + 4 * j + 3] : 0u;
            }
        }
    };
    auto store_piece = [&](const uint4 (&v)[NV]) {
        const uint32_t r0 = (uint32_t)RW - (flw & (uint32_t)(RW - 1));
        uint32_t *dst = myring + r0 * 64u;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            dst[-(4 * j + 0) * 64] = v[j].x; dst[-(4 * j + 1) * 64] = v[j].y;
            dst[-(4 * j + 2) * 64] = v[j].z; dst[-(4 * j + 3) * 64] = v[j].w;
        }
        if (r0 == (uint32_t)RW) {
            myring[0] = v[0].x;
            if (PAIR) myring[-64] = v[0].y;
        }
        flw += (uint32_t)LW;
    };

    uint32_t Q = 0u - 32u * s0;  // minus the bit position (relative to A)
    // Q_need: position (in Q units, Q decreases) at which this lane must have its next piece;
    // 0x7fffffff away from Q means "never" (stream exhausted)
    uint32_t Q_need;
    auto set_limits = [&]() __attribute__((always_inline)) {
        // avail = flw - cw < X  <=>  cw > flw - X  <=>  Q <= ~(32 * (flw - X + 1) - 1) ... kept simple:
        // cw = (~Q) >> 5, so cw >= c  <=>  ~Q >= 32 c  <=>  Q <= ~(32 c)
        const bool more = flw < endw;
        Q_need = more ? ~(32u * (flw - NEED_AT + 1u)) : Q - 0x7fffffffu;
    };
    auto sync_refill = [&]() __attribute__((always_inline)) {  // serve every lane that is (nearly) dry, waiting for the data
        for (;;) {
            const uint32_t avail = (flw - ((~Q) >> 5)) & WMASK;
            const bool more = flw < endw;
            if (!__any(more && avail < NEED_AT)) break;
            if (more && avail <= (uint32_t)(RW - LW) && !(kAblate && (G.dbg & 2u))) {
                uint4 v[NV];
                load_piece(v);
                store_piece(v);
            } else if (more && avail <= (uint32_t)(RW - LW)) {
                flw += (uint32_t)LW;
            }
            wave_sync();
        }
    };

    {   // start-up: the piece that holds word s0 and as many more as fit
        uint4 v[NV];
        for (int i = 0; i < RW / LW; ++i) {
            if (__ballot(flw < endw && flw + (uint32_t)LW <= s0 + (uint32_t)RW) == 0) break;
            if (flw < endw && flw + (uint32_t)LW <= s0 + (uint32_t)RW) {
                if (!(kAblate && (G.dbg & 2u))) { load_piece(v); store_piece(v); } else flw += (uint32_t)LW;
            }
        }
        wave_sync();
    }
    int32_t acc = 0;
    // GEN: inverse of a general prediction filter with taps[0] = +-1 and at most 4 taps (src/deltaRice.c:92-101):
    // y[i] = +-(d[i] - sum_{j=1..3} taps[j] y[i-j]) in int16, i.e. modulo 2^16 (only the low 16 bits of the
    // products count, so 24-bit multiplies of whatever the registers hold above bit 15 are exact).
    // acc is y[i-1]; y2, y3 the two before it; all zero before the waveform (:96 `if ((i - j) >= 0)`).
    int32_t y2 = 0, y3 = 0;
    const uint32_t nt1 = GEN ? G.fast_nt[0] & 0xffffu : 0u, nt2 = GEN ? G.fast_nt[1] & 0xffffu : 0u,
                   nt3 = GEN ? G.fast_nt[2] & 0xffffu : 0u;
    const bool t0neg = GEN && G.fast_t0neg;
    auto advance = [&](int32_t d) __attribute__((always_inline)) {
        if constexpr (GEN) {
            uint32_t a = (uint32_t)d + __umul24(nt1, (uint32_t)acc) + __umul24(nt2, (uint32_t)y2) + __umul24(nt3, (uint32_t)y3);
            if (t0neg) a = 0u - a;
            y3 = y2;
            y2 = acc;
            acc = (int32_t)a;
        } else {
            acc += d;
        }
    };

    // Where the lane's last code ended (as Q): a valid waveform's codes end inside its last payload word, i.e.
    // n_i = ceil(bits / 32) (src/deltaRice.c:237-241) -- checked after the last round, so that flipped payload bits
    // that change a code length are reported (DRX_ERR_CORRUPT) instead of decoding to garbage silently.  The last
    // step of a lane lies in an edge round or is the last step of an interior round: only those capture.
    const uint32_t hi_step = phi + len;  // this lane decodes its last sample in step hi_step - 1
    uint32_t Q_end = 0;
    uint32_t stg[STG ? T / 2 : 1];  // STG: the round's samples of this lane, two per dword (interior rounds)
    auto decode_group = [&](auto first_tag, auto edge_tag, auto stg_tag, int tg, uint32_t tcur) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_tag)::value;
        constexpr bool EDGE = decltype(edge_tag)::value;  // tcur + u is the step index; capture Q_end
        constexpr int TGC = decltype(stg_tag)::value;     // >= 0: tg at compile time, samples go to stg[] (interior rounds of STG)
        if (PAIR && !FIRST) {
            // two samples per ring access: a 64-bit window (three words) always holds two codes (2 x 25 bits),
            // so the second sample's window is one v_alignbit away from the first one's length -- one LDS
            // round trip on the dependent chain per two samples instead of one per sample
#pragma unroll
            for (int u = 0; u < GS; u += 2) {
                const uint32_t row = __builtin_amdgcn_ubfe(Q, 5u, (uint32_t)LOG_RW);
                const uint32_t *wp = myring + row * 64u;
                const uint32_t lo = wp[0], hi = wp[64], lo2 = wp[-64];
                const uint32_t winA = __builtin_amdgcn_alignbit(hi, lo, Q);
                const uint32_t winB = __builtin_amdgcn_alignbit(lo, lo2, Q);
                const uint32_t q1 = ffbh(winA);
                const uint32_t kk1 = (winA < (1u << 24)) ? 16u : k;
                const uint32_t nu1 = ~(q1 + kk1);  // minus the code length
                const uint32_t win2 = __builtin_amdgcn_alignbit(winA, winB, nu1);
                const uint32_t q2 = ffbh(win2);
                const uint32_t kk2 = (win2 < (1u << 24)) ? 16u : k;
                const uint32_t nu2 = ~(q2 + kk2);
                // v_bfe_u32 and v_alignbit_b32 read 5 bits of their offset / shift: ~t == 31 - t (mod 32) serves both
                if constexpr (EDGE) {
                    Q_end = (tcur + (uint32_t)u + 1u == hi_step) ? Q + nu1 : Q_end;
                    Q_end = (tcur + (uint32_t)u + 2u == hi_step) ? Q + nu1 + nu2 : Q_end;
                }
                asm("v_add3_u32 %0, %1, %2, %3" : "=v"(Q) : "v"(Q), "v"(nu1), "v"(nu2));
                const uint32_t z1 = (q1 << kk1) + __builtin_amdgcn_ubfe(winA, nu1, kk1);
                const uint32_t z2 = (q2 << kk2) + __builtin_amdgcn_ubfe(win2, nu2, kk2);
                advance((int32_t)(z1 >> 1) ^ -(int32_t)(z1 & 1u));
                const uint32_t a1 = (uint32_t)acc;
                advance((int32_t)(z2 >> 1) ^ -(int32_t)(z2 & 1u));
                // low halves of the two running sums in one v_perm_b32
                if constexpr (TGC >= 0)
                    stg[(TGC + u) / 2] = __builtin_amdgcn_perm((uint32_t)acc, a1, 0x05040100u);
                else
                    *reinterpret_cast<uint32_t *>(myout + DRX_OIDX(tg + u)) = __builtin_amdgcn_perm((uint32_t)acc, a1, 0x05040100u);
            }
            return;
        }
#pragma unroll
        for (int u = 0; u < GS; ++u) {
            const uint32_t row = __builtin_amdgcn_ubfe(Q, 5u, (uint32_t)LOG_RW);
            const uint32_t *wp = myring + row * 64u;
            const uint32_t lo = wp[0], hi = wp[64];
            const uint32_t win = __builtin_amdgcn_alignbit(hi, lo, Q);
            const uint32_t q = ffbh(win);  // win == 0 only past the end of a corrupt stream
            const bool esc = win < (1u << 24);
            const uint32_t kk = esc ? 16u : k;
            const uint32_t used = q + kk + 1u;
            const uint32_t rem = __builtin_amdgcn_ubfe(win, 32u - used, kk);
            const uint32_t z = (q << kk) + rem;  // escape: 8 << 16 stays above bit 15
            const int32_t d = (int32_t)(z >> 1) ^ -(int32_t)(z & 1u);
            if (FIRST) {
                const bool act = (uint32_t)(tg + u) >= phi;
                const int32_t o1 = acc, o2 = y2, o3 = y3;
                advance(d);
                acc = act ? acc : o1;
                y2 = act ? y2 : o2;
                y3 = act ? y3 : o3;
                Q = act ? Q - used : Q;
            } else {
                advance(d);
                Q -= used;
            }
            if constexpr (EDGE) Q_end = (tcur + (uint32_t)u + 1u == hi_step) ? Q : Q_end;
            if constexpr (TGC >= 0) {  // (two steps fill a staging dword)
                if ((u & 1) == 0) stg[(TGC + u) / 2] = (uint32_t)acc & 0xffffu;
                else stg[(TGC + u) / 2] |= (uint32_t)acc << 16;
            } else {
                myout[DRX_OIDX(tg + u)] = (uint16_t)acc;
            }
        }
    };

    // piece i of an EDGE round (first / last rounds of a waveform): only the samples that belong to the stream
    auto emit_masked = [&](g_i16 *dst, uint32_t lo, uint32_t hi, uint32_t t0, const uint4 &v) __attribute__((always_inline)) {
        const int p = lane % PPS;
        const uint32_t tpos = t0 + 8u * (uint32_t)p;  // step index of the piece's first sample
        if (tpos + 8u > lo && tpos < hi) {
            if (tpos >= lo && tpos + 8u <= hi) {
                *(g_uint4 *)dst = (u32x4v){v.x, v.y, v.z, v.w};
            } else {
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (tpos + (uint32_t)j >= lo && tpos + (uint32_t)j < hi)
                        dst[j] = (int16_t)(w[j >> 1] >> (16 * (j & 1)));
            }
        }
    };
    auto write_out = [&](uint32_t t0) __attribute__((always_inline)) {
        if (kNoObuf || (kAblate && (G.dbg & 1u))) return;
        if (t0 >= lo_max && t0 + T <= hi_min) {  // interior round: whole aligned lines only
#pragma unroll
            for (int i = 0; i < PPS; ++i) {
                const int st = i * SPI + lane / PPS, p = lane % PPS;
                const uint4 v = orow16(st, p);
                store16(outg + wo_off[i] + t0, v);
            }
        } else {
#pragma unroll
            for (int i = 0; i < PPS; ++i) {
                const int st = i * SPI + lane / PPS, p = lane % PPS;
                const uint32_t tpos = t0 + 8u * (uint32_t)p;
                if (tpos + 8u > wo_lo[i] && tpos < wo_hi[i]) emit_masked(outg + wo_off[i] + t0, wo_lo[i], wo_hi[i], t0, orow16(st, p));
            }
        }
    };
    // STG: the round's samples change hands through the workgroup's buffer: rows of 128 bytes, block j of row r at block
    // j ^ (r & 7) (conflict-free both ways); nothing but these sixteen LDS operations happens under the lock
    auto flush_exchange = [&](uint4 (&ov)[PPS]) __attribute__((always_inline)) {
        if constexpr (STG) {
            const uint32_t rowx = (uint32_t)lane * 32u + (uint32_t)(lane & 7) * 4u;
            const uint32_t rd = (uint32_t)(lane >> 3) * 32u + 4u * (uint32_t)((lane & 7) ^ (lane >> 3));
            flush_lock();
#pragma unroll
            for (int j = 0; j < T / 8; ++j)
                *reinterpret_cast<uint4 *>(obuf + (rowx ^ (4u * (uint32_t)j))) = make_uint4(stg[4 * j], stg[4 * j + 1], stg[4 * j + 2], stg[4 * j + 3]);
            wave_sync();
#pragma unroll
            for (int i = 0; i < PPS; ++i) ov[i] = *reinterpret_cast<const uint4 *>(obuf + (uint32_t)i * 256u + rd);
            flush_unlock();
        }
    };
    // the samples of the group that ends here are final (see the interior loop)
    auto pin_group = [&](auto gi) __attribute__((always_inline)) {
        if constexpr (STG) {
            constexpr int TG = decltype(gi)::value * GS;
            int32_t &a_ = acc;  // (named here so that the generic lambda captures them: asm operands alone do not)
            uint32_t (&s_)[STG ? T / 2 : 1] = stg;
            asm volatile("" : "+v"(a_), "+v"(s_[TG / 2]), "+v"(s_[TG / 2 + 1]), "+v"(s_[TG / 2 + 2]), "+v"(s_[TG / 2 + 3]),
                         "+v"(s_[TG / 2 + 4]), "+v"(s_[TG / 2 + 5]), "+v"(s_[TG / 2 + 6]), "+v"(s_[TG / 2 + 7]));
        }
    };
    auto stg_line = [&](int i, uint32_t t0) __attribute__((always_inline)) -> g_i16 * {  // piece i of this lane in round t0
        return (outg + line_base + t0) + wo_rel[i];
    };
    auto stg_edge_out = [&](uint32_t t0) __attribute__((always_inline)) {  // STG: an edge round's masked write-out
        uint4 ov[PPS];
        flush_exchange(ov);
        if (kAblate && (G.dbg & 1u)) return;
        if (t0 >= lo_max && t0 + T <= hi_min) {
#pragma unroll
            for (int i = 0; i < PPS; ++i) store16(stg_line(i, t0), ov[i]);
        } else {
#pragma unroll
            for (int i = 0; i < PPS; ++i) {  // (the stream's limits come by shuffle: edge rounds are 2-3 of ~110)
                const int st = i * SPI + lane / PPS;
                emit_masked(stg_line(i, t0), (uint32_t)__shfl((int)phi, st), (uint32_t)__shfl((int)(phi + len), st), t0, ov[i]);
            }
        }
    };

    // round 0 (start delays; synchronous refills; runs once)
    if (steps > 0) {
        if constexpr (STG) {
            static_for<0, T / GS>([&](auto gi) __attribute__((always_inline)) {
                constexpr int TG = decltype(gi)::value * GS;
                sync_refill();
                decode_group(std::true_type{}, std::true_type{}, std::integral_constant<int, TG>{}, TG, (uint32_t)TG);
                pin_group(gi);
            });
            wave_sync();
            stg_edge_out(0);
        } else {
#pragma unroll 1
            for (int tg = 0; tg < T; tg += GS) {
                sync_refill();
                decode_group(std::true_type{}, std::true_type{}, std::integral_constant<int, -1>{}, tg, (uint32_t)tg);
            }
            wave_sync();
            write_out(0);
        }
        wave_sync();
    }

    // steady state.  Stream pieces are requested at the END of a round, just BEFORE the round's stores
    // are issued, and written to the ring at the end of the next round.  vmcnt retires in issue order
    // and counts loads and stores together, so a wait for loads that are OLDER than the PPS stores of
    // their own round is `s_waitcnt vmcnt(PPS)` and never waits for those stores (ablation: loads alone
    // +0.04 ms, stores alone +0.11 ms, both +1.0 ms when the commit had to drain the stores too).
    // The interior rounds (every stream fully inside its waveform: whole-line stores only) run in a
    // loop of their own whose only vector-memory operations are those loads and those PPS stores, so
    // that the compiler's waitcnt insertion can prove the count.  Up to two pieces per lane and round
    // (2 LW words = 16 bits per sample at LW = 16); hungrier streams fall back to sync_refill().
    auto edge_round = [&](uint32_t t0) __attribute__((always_inline)) {  // first / last rounds: masked stores, synchronous refills
        if constexpr (STG) {
            static_for<0, T / GS>([&](auto gi) __attribute__((always_inline)) {
                constexpr int TG = decltype(gi)::value * GS;
                sync_refill();
                decode_group(std::false_type{}, std::true_type{}, std::integral_constant<int, TG>{}, TG, t0 + (uint32_t)TG);
                pin_group(gi);
            });
            wave_sync();
            stg_edge_out(t0);
        } else {
#pragma unroll 1
            for (int tg = 0; tg < T; tg += GS) {
                sync_refill();
                decode_group(std::false_type{}, std::true_type{}, std::integral_constant<int, -1>{}, tg, t0 + (uint32_t)tg);
            }
            wave_sync();
            write_out(t0);
        }
        wave_sync();
    };
    uint32_t t0 = T;
    for (; t0 < steps && !(t0 >= lo_max && t0 + T <= hi_min); t0 += T) edge_round(t0);

    const uint32_t min_words = (kAblate && (G.dbg & 4u)) ? 0u : ((uint32_t)T * (k + 1u)) >> 5;
    uint4 pv0[NV], pv1[NV];               // pieces in flight
    bool pneed0 = false, pneed1 = false;  // this lane has them in flight
    set_limits();
    auto group_refill = [&]() __attribute__((always_inline)) {
        if (__any((int32_t)(Q - Q_need) <= 0)) {  // one signed compare per test (positions are mod 2^32)
            // a piece in flight was requested counting on the words this round consumes at least
            // (min_words): before the round is over it may only be committed where it already fits
            uint32_t avail = (flw - ((~Q) >> 5)) & WMASK;
            if (pneed0 && avail <= (uint32_t)(RW - LW)) {
                store_piece(pv0);
                pneed0 = false;
                avail += (uint32_t)LW;
                if (pneed1 && avail <= (uint32_t)(RW - LW)) { store_piece(pv1); pneed1 = false; }
            }
            if (!pneed0 && pneed1) {
#pragma unroll
                for (int j = 0; j < NV; ++j) pv0[j] = pv1[j];
                pneed0 = true;
                pneed1 = false;
            }
            wave_sync();
            sync_refill();
            set_limits();
        }
    };
    for (; t0 + T <= hi_min && t0 < steps; t0 += T) {  // interior rounds
        if constexpr (STG) {  // fully unrolled: the staging registers are indexed at compile time
            static_for<0, T / GS>([&](auto gi) __attribute__((always_inline)) {
                constexpr int TG = decltype(gi)::value * GS;
                group_refill();
                decode_group(std::false_type{}, std::false_type{}, std::integral_constant<int, TG>{}, TG, 0u);
                // the group's values are final HERE: without this the compiler runs the bit-position chain of the whole
                // unrolled round ahead (the next group's refill test needs only that) and parks every window, quotient and
                // width of sixteen samples in registers until it gets round to the values: 415 VGPRs and 350 spills
                pin_group(gi);
            });
        } else {
#pragma unroll 1
            for (int tg = 0; tg < T; tg += GS) {
                group_refill();
                decode_group(std::false_type{}, std::false_type{}, std::integral_constant<int, -1>{}, tg, 0u);
            }
        }
        Q_end = (t0 + (uint32_t)T == hi_step) ? Q : Q_end;  // a lane whose last step closes an interior round
        wave_sync();
        if (pneed0) store_piece(pv0);  // loads of the previous round end: older than that round's PPS stores
        if (pneed1) store_piece(pv1);
        wave_sync();
        set_limits();
        {
            // a piece is committed one interior round after its request: by then every lane has decoded T more
            // samples of at least k + 1 bits each, i.e. consumed min_words more words
            const uint32_t mc = (t0 + 2u * T <= hi_min && t0 + T < steps) ? min_words : 0u;
            const uint32_t avail = (flw - ((~Q) >> 5)) & WMASK;
            pneed0 = (flw < endw) && avail + (uint32_t)LW <= (uint32_t)RW + mc && !(kAblate && (G.dbg & 2u));
            pneed1 = pneed0 && (flw + (uint32_t)LW < endw) && avail + 2u * (uint32_t)LW <= (uint32_t)RW + mc;
            if (pneed0) load_piece(pv0);
            if (pneed1) load_piece(pv1, (uint32_t)LW);
        }
        if constexpr (STG) {
            uint4 ov[PPS];
            flush_exchange(ov);
            if (!(kAblate && (G.dbg & 1u))) {
#pragma unroll
                for (int i = 0; i < PPS; ++i) store16(stg_line(i, t0), ov[i]);
            }
        } else if (!(kNoObuf || (kAblate && (G.dbg & 1u)))) {
#pragma unroll
            for (int i = 0; i < PPS; ++i) {  // whole aligned lines only
                const int st = i * SPI + lane / PPS, p = lane % PPS;
                const uint4 v = orow16(st, p);
                store16(outg + wo_off[i] + t0, v);
            }
        }
        wave_sync();
    }
    if (pneed0) store_piece(pv0);
    if (pneed1) store_piece(pv1);
    wave_sync();
    for (; t0 < steps; t0 += T) edge_round(t0);
    // bits of the waveform = -Q_end - 32 s0 (Q counts from A); n_i words hold them exactly
    if (active && len && (((0u - Q_end) - 32u * s0 + 31u) >> 5) != n) atomicOr(&st->err, kErrCorrupt);
}

// ---------------------------------------------------------------------------
// launchers (host side, same translation unit so that <<<>>> stays in HIP code)
// ---------------------------------------------------------------------------
hipError_t launch_sideband_tables(const Geom &G, const uint32_t *d_in, uint64_t in_words, const uint64_t *d_chunk_word_off,
                                  const uint32_t *d_n, uint64_t *d_wave_off, uint32_t *d_wave_words, DevStatus *d_status, hipStream_t s) {
    if (G.total_waves == 0) return hipSuccess;
    k_sideband_tables<<<(unsigned)G.n_chunks, 256, 0, s>>>(G, d_in, in_words, d_chunk_word_off, d_n, d_wave_off, d_wave_words, d_status);
    return hipGetLastError();
}

uint32_t bw_walk_blocks_max(const Geom &G) {
    // against the walk inside the decode launch (serial chase through LDS, 0.13 us per waveform of a chunk, all chunks at
    // once, so that large batches hide most of it).  Measured (chunks of 14 M samples, decode GB/s of the two paths at
    // 150 / 220 chunks): L = 512 1602 / 1627 against 875 / 1265; L = 1024 1869 / 1927 against 1521 / 1954; L = 2048
    // 1906 / 2012 against 1673 / 2169: about one chunk per 35 waveforms of a chunk.  Above WaveformLength 2048 the
    // alternative is the scalar chain at 0.85 us per hop (L = 3072, 100 / 220 chunks: 1731 / 1459+ against 596 / 1123): W / 18
    const uint64_t per = G.u_wave_len <= kWalkShortLen ? 35u : 18u, cap = kPwMaxChunks;
    const uint64_t limit = G.u_n_waves / per < cap ? G.u_n_waves / per : cap;
    // every 4096-word block must hold a header: n_i <= 25 L / 32 < 4096, i.e. L <= 5000; chunks of longer
    // waveforms within the chunk-wide walk's capacity take that one
    const bool chunk_wide = G.u_wave_len > kWalkShortLen && G.u_n_waves <= kSwMaxWaves && G.u_n_waves >= kSwMinWaves;
    if (!(G.uniform && G.n_chunks <= limit && G.u_wave_len <= 5000u && G.u_wave_len >= 16u && !chunk_wide)) return 0;
    const uint64_t max_words = 1u + G.u_n_waves + (((uint64_t)G.u_n_samples * 25u + 31u) >> 5) + G.u_n_waves;
    const uint64_t nb = (max_words + kWalkBlockWords - 1u) / kWalkBlockWords;
    return nb > 0xfffffu ? 0u : (uint32_t)nb;
}

// bytes of header list per 4096 words of stream for the block size the launch will choose (0: none kept)
static uint64_t bw_hop_bytes_per_block4096(const Geom &G) {
    const uint32_t max_len = G.uniform ? G.u_wave_len : kWalkShortLen, min_len = G.uniform ? G.u_wave_len : G.rag_bw_min_len;
    const uint32_t max_full = (uint32_t)(((uint64_t)max_len * 25u + 31u) >> 5);
    const uint32_t B = max_full + 2u <= 1024u ? 1024u : (max_full + 2u <= 2048u ? 2048u : 4096u);
    return (uint64_t)bw_hop_cap(B, min_len, G.k) * sizeof(uint32_t) * (kWalkBlockWords / B);
}

// scratch of the parallel header walks (0: the batch takes neither); layout in launch_decode()
uint64_t par_walk_scratch_bytes(const Geom &G) {
    bool pw, bw;
    uint64_t bw_units;
    if (G.uniform) {
        const uint32_t nb = bw_walk_blocks_max(G);
        bw = nb != 0;
        bw_units = G.n_chunks * nb;
        pw = !bw && G.n_chunks <= kSwMaxChunks && G.u_n_waves <= kSwMaxWaves && G.u_n_waves >= kSwMinWaves && G.u_wave_len > kWalkShortLen;
    } else {
        if (!G.rag_par) return 0;
        pw = G.n_long != 0;
        bw = G.n_short != 0;
        bw_units = (uint64_t)G.n_short * G.rag_bw_blocks_max;
    }
    if (!pw && !bw) return 0;
    // (the candidate lists of the scan form only where that form may run)
    return (pw && G.n_chunks <= kPwMaxChunks ? G.n_chunks * kPwStride * sizeof(uint2) : 0) + (3u * G.n_chunks + 2u) * sizeof(uint32_t) +
           bw_units * (kWalkBlockWords / 1024u) * sizeof(BwBlock) +  // (blocks of 1024 words at the smallest)
           bw_units * bw_hop_bytes_per_block4096(G);                 // header lists of the first block pass
}

// Ablation builds only: DRX_DEC_LDS_PAD = bytes of dynamic LDS added to every k_decode_lanes launch (occupancy A/B at an
// unchanged instruction stream: 26 KB + pad per wavefront decides how many of them a CU holds).
static unsigned dec_lds_pad() {
#ifdef DRX_ABLATION
    static const unsigned pad = [] { const char *e = getenv("DRX_DEC_LDS_PAD"); return e ? (unsigned)atoi(e) : 0u; }();
    return pad;
#else
    return 0u;
#endif
}

#ifndef DRX_DEC_T
#define DRX_DEC_T 64
#endif
#ifndef DRX_DEC_LW
#define DRX_DEC_LW 16
#endif
// ---------------------------------------------------------------------------
// How a decode call is routed: ONE table, first matching row wins (route_decode()).
//
//   decoder       | when                                                                              | walk in front of it
//   --------------+-----------------------------------------------------------------------------------+--------------------------------
//   SIMPLE        | decode_impl 0; a filter the fast kernels do not take (> 4 taps, taps[0] != +-1)   | parallel walks / serial
//   BLOCKS (+IIR) | few long waveforms (blocks_batch(): geometry and cost), delta or a fast filter    | parallel walks / serial
//   LONG          | uniform, delta, long_waveform_batch() and not BLOCKS; flag 512                    | parallel walks / serial
//   LANES fused   | decode_impl 8 (5), not a batch the parallel walks take (short waveforms in many    | inside the launch
//                 | chunks, chunks of more than 8192 or fewer than 8 waveforms), grid not mostly idle |
//   LANES         | everything else; ragged batches behind both parallel walks: two launches          | parallel walks / serial
//
//   walk          | when (never with tables_ready: the caller filled wave_off / wave_words)
//   --------------+-----------------------------------------------------------------------------------
//   chunk-wide    | chunks of 8 ... 8192 waveforms longer than 2048 samples (uniform, any number of chunks: k_walk_sparse chases 64 chains per
//                 | chunk without reading it -- the headline batch too), the long-waveform chunks of a small ragged batch
//   block-parallel| bw_walk_blocks_max(): few chunks of many short waveforms (uniform), the short-waveform chunks of a small ragged batch
//   serial        | otherwise: LDS block walkers (WaveformLength <= 2048) / scalar chains, one launch in front of the decoder
// Flags: 256 never BLOCKS / LONG, 512 LONG instead of BLOCKS, 2048 never the parallel walks, 131072 one lanes launch behind both walks.
// ---------------------------------------------------------------------------
enum class Dec { Simple, Blocks, Long, LanesFused, Lanes };
enum class Walk { None, InLaunch, Parallel, Serial };
struct DecodeRoute {
    Dec dec;
    Walk walk;
    bool pair;            // two samples per ring access (decode_impl 7 / 8; 1 / 5 are legacy builds')
    bool gen;             // a general prediction filter
    bool use_pw, use_bw;  // Walk::Parallel: the chunk-wide walk, the block-parallel walk
    uint32_t bw_blocks_max;
};

static DecodeRoute route_decode(const Geom &G, int impl, bool tables_ready, bool have_pw, bool have_blk) {
    DecodeRoute R{};
    R.gen = G.n_taps != 0;
    const bool simple = impl == 0 || (R.gen && !G.fast_taps);
#ifdef DRX_LEGACY
    R.pair = R.gen ? true : (impl == 7 || impl == 8);  // (general filters exist in the two-samples form only)
#else
    R.pair = true;
#endif
    const bool want_fused = !tables_ready && (impl == 5 || impl == 8);
    const bool no_par = tables_ready || !have_pw || (G.dbg & 2048u);
    const bool par_walk = !no_par && G.uniform && G.n_chunks <= kSwMaxChunks && G.u_n_waves <= kSwMaxWaves && G.u_n_waves >= kSwMinWaves &&
                          G.u_wave_len > kWalkShortLen;
    R.bw_blocks_max = bw_walk_blocks_max(G);
    const bool bw_walk = !no_par && R.bw_blocks_max != 0;
    const bool rag_par = !no_par && !G.uniform && G.rag_par;
    R.use_pw = par_walk || (rag_par && G.n_long);
    R.use_bw = bw_walk || (rag_par && G.n_short);
    const bool blocks = !simple && !(G.dbg & (256u | 512u)) && have_blk && blocks_batch(G) && (!R.gen || (G.iir_tab && G.iir_state));
    const bool longp = !simple && !blocks && !R.gen && !(G.dbg & 256u) && G.uniform && long_waveform_batch(G.total_waves, G.u_wave_len);
    // ragged: the group-major grid of the fused launch has max_groups tickets per chunk; not when most of them would be idle
    const bool sparse = !G.uniform && (uint64_t)G.n_chunks * G.max_groups > 8ull * ((G.total_waves + 63u) / 64u) + 4096ull;
    R.dec = simple ? Dec::Simple : (blocks ? Dec::Blocks : (longp ? Dec::Long : Dec::Lanes));
    const bool parallel = par_walk || bw_walk || rag_par;
    if (R.dec == Dec::Lanes && want_fused && !sparse && !parallel) R.dec = Dec::LanesFused;
    R.walk = tables_ready ? Walk::None : (R.dec == Dec::LanesFused ? Walk::InLaunch : (parallel ? Walk::Parallel : Walk::Serial));
    return R;
}

hipError_t launch_decode(const Geom &G, const uint32_t *d_in, uint64_t in_words,
                         const uint64_t *d_chunk_word_off, int16_t *d_out, uint64_t *d_wave_off,
                         uint32_t *d_wave_words, uint64_t *d_granules, DevStatus *d_status, int impl,
                         void *d_pw, void *d_blk, const SideStream *side, hipEvent_t *ev, hipStream_t s, uint32_t *path_out) {
    uint32_t path_dummy = 0;
    uint32_t &path = path_out ? *path_out : path_dummy;
    path = 0;
    if (G.total_waves == 0) return hipSuccess;
    const unsigned lpad = dec_lds_pad();
    mark(ev, 0, s);
    // impl >= 100: wave_off / wave_words are already filled in (the one-chunk host path walks the header chain on the CPU while
    // the chunk is in flight to the device; the side-band decode derives them from the caller's table): no walk
    const bool tables_ready = impl >= 100;
    if (tables_ready) impl -= 100;
    const DecodeRoute R = route_decode(G, impl, tables_ready, d_pw != nullptr, d_blk != nullptr);
    const bool gen = R.gen;
    const unsigned nb_plain = blocks_for(G.total_waves, 64);

    // the lane-per-waveform launch behind a walk (tables in wave_off / wave_words): `nb` wavefronts of view Gv
    auto launch_lanes = [&](const Geom &Gv, unsigned nb, hipStream_t st_) {
        path |= 2u;  // DRX_PATH_LANES
        if (gen)
            k_decode_lanes<64, DRX_DEC_LW, DRX_DEC_T, 16, false, true, true><<<nb, 64, lpad, st_>>>(Gv, d_in, in_words, d_chunk_word_off, d_wave_off, d_wave_words, nullptr, nullptr, d_status, d_out);
        else if (R.pair)
            k_decode_lanes<64, DRX_DEC_LW, DRX_DEC_T, 16, false, true><<<nb, 64, lpad, st_>>>(Gv, d_in, in_words, d_chunk_word_off, d_wave_off, d_wave_words, nullptr, nullptr, d_status, d_out);
#ifdef DRX_LEGACY
        else
            k_decode_lanes<64, DRX_DEC_LW, DRX_DEC_T, 16, false><<<nb, 64, lpad, st_>>>(Gv, d_in, in_words, d_chunk_word_off, d_wave_off, d_wave_words, nullptr, nullptr, d_status, d_out);
#endif
    };

    // ---- the walk ----
    bool lanes_done = false;
    if (R.walk == Walk::InLaunch) {
        // granules + ticket word, zeroed before every launch (a granule is its own ready flag)
        hipError_t e = hipMemsetAsync(d_granules, 0, (G.total_waves + 2) * sizeof(uint64_t), s);
        if (e != hipSuccess) return e;
    } else if (R.walk == Walk::Parallel) {
        // scratch: uint2 cand[n_chunks * kPwCap] | uint32 count[n_chunks] | pw_fail[n_chunks] | bw_fail[n_chunks] |
        //          BwBlock info[n_bw * bw_blocks]   (cand .. pw_fail only where the chunk-wide walk is used)
        const bool use_pw = R.use_pw, use_bw = R.use_bw;
        const uint32_t *pw_list = G.uniform ? nullptr : G.walk_long, *bw_list = G.uniform ? nullptr : G.walk_short;
        const uint32_t n_pw = G.uniform ? (uint32_t)G.n_chunks : G.n_long, n_bw = G.uniform ? (uint32_t)G.n_chunks : G.n_short;
        const uint32_t bwb = G.uniform ? R.bw_blocks_max : G.rag_bw_blocks_max;
        uint2 *cand = reinterpret_cast<uint2 *>(d_pw);
        const bool have_cand = use_pw && G.n_chunks <= kPwMaxChunks;  // (par_walk_scratch_bytes())
        uint32_t *cnt = reinterpret_cast<uint32_t *>(cand + (have_cand ? G.n_chunks * kPwStride : 0));
        uint32_t *pw_fail = cnt + G.n_chunks, *bw_fail = pw_fail + G.n_chunks;
        BwBlock *info = reinterpret_cast<BwBlock *>(bw_fail + G.n_chunks + (G.n_chunks & 1u));
        hipError_t e = hipMemsetAsync(cnt, 0, 3u * G.n_chunks * sizeof(uint32_t), s);
        if (e != hipSuccess) return e;
        // a ragged batch has both kinds of chunk and the two walks touch different chunks: the chunk-wide walk goes to the
        // context's side stream while the block walk runs here (config 5: 0.18 ms of 0.6 off the critical path)
        const bool forked = use_pw && use_bw && side && side->s;
        hipStream_t spw = forked ? side->s : s;
        if (forked) {
            if ((e = hipEventRecord(side->fork, s)) != hipSuccess) return e;
            if ((e = hipStreamWaitEvent(side->s, side->fork, 0)) != hipSuccess) return e;
        }
        if (use_pw) {
            // (up to four chunks the scan form is quicker: 128 workgroups read one chunk in 11 us, where a chain is 31 dependent loads)
            // ... and chunks of a few very long waveforms (ragged batches): a start costs half a waveform's code in reads
            // (debug flag 16777216: chains whatever the batch -- the tests' small batches)
            const bool few_waves = G.uniform && (G.u_n_waves < 64u || G.u_n_waves > kPwMaxWaves);  // (outside the scan form's range)
            const bool chains_suit = (G.dbg & 16777216u) || few_waves || (n_pw > 4u && (G.uniform || G.rag_pw_min_waves >= 64u));
            if ((!(G.dbg & 8388608u) && chains_suit) || !have_cand || few_waves) {
                // 64 chains per chunk chased in parallel from starts found by looking forward from 64 cuts (drx_walk.h): the chunk is
                // not read
                const unsigned sw_threads = (G.uniform && G.u_n_waves < 256u) ? 256u : (unsigned)kSwThreads;  // (few chains: few wavefronts)
                k_walk_sparse<<<n_pw, sw_threads, 0, spw>>>(G, d_in, in_words, d_chunk_word_off, d_wave_off, d_wave_words, pw_fail, pw_list);
            } else {
                k_pw_scan<<<(unsigned)(n_pw * pw_parts(n_pw)), 256, 0, spw>>>(G, d_in, in_words, d_chunk_word_off, pw_list, cand, cnt, pw_parts(n_pw));
                k_walk_parallel<<<n_pw, kPwThreads, 0, spw>>>(G, d_in, in_words, d_chunk_word_off, d_wave_off, d_wave_words,
                                                              pw_fail, pw_list, cand, cnt, pw_parts(n_pw));
            }
            k_walk_scalar_only<<<blocks_for(G.n_chunks, kWalkChains), 64, 0, spw>>>(G, d_in, in_words, d_chunk_word_off, d_wave_off,
                                                                                    d_wave_words, d_status, pw_fail);
        }
        // ... and so does the decoding of the long-waveform chunks, whose tables are complete long before the block walk is
        // through: their wavefronts (the first rag_groups_long of the longest-first order) are launched behind the
        // chunk-wide walk on the side stream, the rest here behind the block walk
        const bool split = forked && R.dec == Dec::Lanes && G.rag_order && G.rag_groups_long && G.rag_groups_long < G.rag_groups &&
                           !(G.dbg & 131072u);
        if (split) {
            Geom Gl = G;
            Gl.rag_groups = G.rag_groups_long;
            launch_lanes(Gl, Gl.rag_groups, spw);
        }
        if (forked && (e = hipEventRecord(side->join, side->s)) != hipSuccess) {
            // the side stream's kernels write this call's tables and output: never return with them unordered
            (void)hipStreamSynchronize(side->s);
            return e;
        }
        if (use_bw) {
            // block size: the smallest that exceeds every listed chunk's max_words; wavefronts: what the LDS lets the chip hold
            const uint32_t max_len = G.uniform ? G.u_wave_len : kWalkShortLen;
            const uint32_t max_full = (uint32_t)(((uint64_t)max_len * 25u + 31u) >> 5);
            const uint32_t min_len = G.uniform ? G.u_wave_len : G.rag_bw_min_len;
            auto run_bw = [&](auto btag, unsigned waves_per_cu) {
                constexpr uint32_t B = decltype(btag)::value;
                const uint32_t bmax = bwb * (kWalkBlockWords / B);
                const unsigned grid = 256u * waves_per_cu;
                // header lists behind info[] (sized for the smallest block: par_walk_scratch_bytes())
                const uint32_t hop_cap = bw_hop_cap(B, min_len, G.k);
                uint32_t *hops = hop_cap ? reinterpret_cast<uint32_t *>(info + (uint64_t)n_bw * bwb * (kWalkBlockWords / 1024u)) : nullptr;
                if (hops)
                    k_bw_blocks<B, false, true><<<grid, 64, 0, s>>>(G, d_in, in_words, d_chunk_word_off, bw_list, n_bw, bmax, info, nullptr, nullptr, nullptr, d_status, hops, hop_cap);
                else
                    k_bw_blocks<B, false><<<grid, 64, 0, s>>>(G, d_in, in_words, d_chunk_word_off, bw_list, n_bw, bmax, info, nullptr, nullptr, nullptr, d_status, nullptr, 0u);
                k_bw_scan<B><<<n_bw, 64, 0, s>>>(G, d_in, in_words, d_chunk_word_off, bw_list, n_bw, bmax, info, bw_fail);
                if (hops)
                    k_bw_emit<B><<<256u * 8u, 256, 0, s>>>(G, in_words, d_chunk_word_off, bw_list, n_bw, bmax, info, bw_fail, hops, hop_cap, d_wave_off, d_wave_words, d_status);
                else
                    k_bw_blocks<B, true><<<grid, 64, 0, s>>>(G, d_in, in_words, d_chunk_word_off, bw_list, n_bw, bmax, info, bw_fail, d_wave_off, d_wave_words, d_status, nullptr, 0u);
            };
            if (max_full + 2u <= 1024u) run_bw(std::integral_constant<uint32_t, 1024>{}, 24u);
            else if (max_full + 2u <= 2048u) run_bw(std::integral_constant<uint32_t, 2048>{}, 13u);
            else run_bw(std::integral_constant<uint32_t, 4096>{}, 7u);
            k_walk_block_only<<<(unsigned)G.n_chunks, 64, 0, s>>>(G, d_in, in_words, d_chunk_word_off, d_wave_off, d_wave_words, d_status, bw_fail);
        }
        if (split) {
            mark(ev, 1, s);  // (the block walk's end; the other stream is decoding already)
            Geom Gs = G;
            Gs.rag_order = G.rag_order + G.rag_groups_long;
            Gs.rag_groups = G.rag_groups - G.rag_groups_long;
            launch_lanes(Gs, Gs.rag_groups, s);
            lanes_done = true;
        }
        if (forked && (e = hipStreamWaitEvent(s, side->join, 0)) != hipSuccess) {
            (void)hipStreamSynchronize(side->s);
            return e;
        }
    } else if (R.walk == Walk::Serial) {
        // chunks of short waveforms: stream the chunk through LDS; long waveforms: one dependent load per hop
        if (G.uniform) {
            if (G.u_wave_len <= kWalkShortLen)
                k_walk_block<<<(unsigned)G.n_chunks, 64, 0, s>>>(G, d_in, in_words, d_chunk_word_off, nullptr,
                                                                 (uint32_t)G.n_chunks, d_wave_off, d_wave_words, d_status);
            else
                k_walk_scalar<<<blocks_for(G.n_chunks, kWalkChains), 64, 0, s>>>(G, d_in, in_words, d_chunk_word_off,
                                                                                 d_wave_off, d_wave_words, d_status);
        } else {
            if (G.n_short) k_walk_block<<<G.n_short, 64, 0, s>>>(G, d_in, in_words, d_chunk_word_off, G.walk_short, G.n_short,
                                                                 d_wave_off, d_wave_words, d_status);
            if (G.n_long) k_walk_list<<<blocks_for(G.n_long, 64), 64, 0, s>>>(G, d_in, in_words, d_chunk_word_off, G.walk_long,
                                                                             G.n_long, d_wave_off, d_wave_words, d_status);
        }
    }
    if (!lanes_done) mark(ev, 1, s);

    // ---- the decoder ----
    switch (lanes_done ? Dec::Simple /* nothing left to launch */ : R.dec) {
    case Dec::LanesFused: {
        uint32_t *ticket = reinterpret_cast<uint32_t *>(d_granules + G.total_waves);
        unsigned n_walk, groups;
        unsigned nb;
        if (G.uniform) {
            n_walk = (G.u_wave_len <= kWalkShortLen) ? (unsigned)G.n_chunks : blocks_for(G.n_chunks, kWalkChains);
            groups = (G.u_n_waves + 63u) / 64u;
            // full groups of every chunk, then the chunks' partial last groups, 64 / (waveforms left) chunks to a wavefront
            const uint32_t full = G.u_n_waves >> 6, rem = G.u_n_waves & 63u;
            nb = n_walk + (unsigned)(G.n_chunks * full) + (rem ? (unsigned)((G.n_chunks + 64u / rem - 1u) / (64u / rem)) : 0u);
        } else {
            n_walk = G.n_short + blocks_for(G.n_long, kWalkChains);
            groups = G.max_groups;
            nb = n_walk + (unsigned)(G.n_chunks * groups);
        }
        path |= 1u;  // DRX_PATH_LANES_FUSED
#ifndef DRX_DEC_NW
#define DRX_DEC_NW 1
#endif
        // staged flush (NW wavefronts share one transposition buffer: 8 or 9 wavefronts per CU instead of 6; measured NOT faster,
        // compiled only with -DDRX_DEC_NW=4 / 9); its walker role has no LDS to stream short-waveform chunks through
        constexpr int NWs = DRX_DEC_NW;
        const bool short_walk = G.uniform ? G.u_wave_len <= kWalkShortLen : G.n_short != 0;
        const bool lines_near = G.uniform ? (uint64_t)G.u_wave_len * 64u < (1ull << 31) : G.max_wave_len64 < (1ull << 31);
        if (NWs > 1 && R.pair && !short_walk && lines_near && !(G.dbg & 1048576u)) {
            const unsigned nwg = (nb + NWs - 1u) / NWs;
            if (gen)
                k_decode_lanes<64, DRX_DEC_LW, 64, 16, true, true, true, NWs><<<nwg, 64 * NWs, lpad, s>>>(G, d_in, in_words, d_chunk_word_off, d_wave_off, d_wave_words, d_granules, ticket, d_status, d_out);
            else
                k_decode_lanes<64, DRX_DEC_LW, 64, 16, true, true, false, NWs><<<nwg, 64 * NWs, lpad, s>>>(G, d_in, in_words, d_chunk_word_off, d_wave_off, d_wave_words, d_granules, ticket, d_status, d_out);
        } else if (gen)
            k_decode_lanes<64, DRX_DEC_LW, DRX_DEC_T, 16, true, true, true><<<nb, 64, lpad, s>>>(G, d_in, in_words, d_chunk_word_off, d_wave_off, d_wave_words, d_granules, ticket, d_status, d_out);
        else if (R.pair)
            k_decode_lanes<64, DRX_DEC_LW, DRX_DEC_T, 16, true, true><<<nb, 64, lpad, s>>>(G, d_in, in_words, d_chunk_word_off, d_wave_off, d_wave_words, d_granules, ticket, d_status, d_out);
#ifdef DRX_LEGACY
        else
            k_decode_lanes<64, DRX_DEC_LW, DRX_DEC_T, 16, true><<<nb, 64, lpad, s>>>(G, d_in, in_words, d_chunk_word_off, d_wave_off, d_wave_words, d_granules, ticket, d_status, d_out);
#endif
        break;
    }
    case Dec::Blocks: {
        // a workgroup per block of every waveform (drx_blocks.hip); waveforms it flags are decoded again, one
        // workgroup each, by the kernel that also judges them
        const uint32_t *fail = nullptr, *suspect = nullptr;
        bool fused = false;
        hipError_t e = launch_decode_blocks(G, d_in, in_words, d_wave_off, d_wave_words, d_blk, d_status, d_out, &fail, &suspect, gen, s, &fused);
        if (e != hipSuccess) return e;
        path |= 4u | (gen ? (fused ? 64u : 32u) : 0u);  // DRX_PATH_BLOCKS (| DRX_PATH_IIR_FUSED / DRX_PATH_IIR)
        k_decode_long<<<(unsigned)G.total_waves, kLongThreads, 0, s>>>(G, d_in, d_wave_off, d_wave_words, d_status, d_out, fail, suspect, gen ? 1u : 0u);
        if (gen) {
            // (not fused: residuals -> samples, in place;) then the waveforms the block decoder flagged, serially (a slope-1 ramp)
            if (!fused && (e = launch_iir(G, G.iir_chunk_tile_base, G.iir_n_tiles, G.iir_tab, G.iir_state, fail, d_status, d_out, s)) != hipSuccess) return e;
            k_decode_simple<<<nb_plain, 64, 0, s>>>(G, d_in, d_wave_off, d_wave_words, d_status, d_out, fail);
        }
        break;
    }
    case Dec::Long:  // (flag 512, or a long-waveform batch the block decoder does not take: one workgroup per waveform)
        path |= 8u;  // DRX_PATH_LONG
        k_decode_long<<<(unsigned)G.total_waves, kLongThreads, 0, s>>>(G, d_in, d_wave_off, d_wave_words, d_status, d_out, nullptr, nullptr, 0u);
        break;
    case Dec::Lanes:
        launch_lanes(G, (!G.uniform && G.rag_order) ? G.rag_groups : nb_plain, s);  // (ragged: groups per chunk round up)
        break;
    case Dec::Simple:
        if (!lanes_done) {
            path |= 16u;  // DRX_PATH_SIMPLE
            k_decode_simple<<<nb_plain, 64, 0, s>>>(G, d_in, d_wave_off, d_wave_words, d_status, d_out, nullptr);
        }
        break;
    }
    mark(ev, 2, s);
    mark(ev, 3, s);
    return hipGetLastError();
}

}  // namespace drx
