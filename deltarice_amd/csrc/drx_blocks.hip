// drx_blocks.hip -- block-parallel decoder: a WORKGROUP per block of a waveform's stream.
//
// The lane-per-waveform decoder (k_decode_lanes, drx_kernels.hip) needs ~10^5 waveforms to fill an MI355X.  The
// reference's own shapes often have far fewer: its default options make every chunk ONE waveform
// (src/deltaRice.c:249-258), nEDM / NOPTREX chunks are 32 waveforms of 81 920 / 500 000 samples
// (docs/Performance.md:27,38), and one H5Z call on the README's example chunk sees 20 waveforms
// (README.md:75-82).  Here the parallelism comes from inside the waveform -- what north_star calls a wavefront per
// waveform, in the only form the format permits: the Rice parse is serial (code boundaries are data dependent,
// src/deltaRice.c:155-171, and the format has no index), but it SELF-SYNCHRONISES: a parse started at an arbitrary
// bit falls into step with the true code boundaries within a few codes.
//
//   block     kWords = NT x 13 words of one waveform's payload, in LDS, reversed word order (word 0 on top) so that
//             with the bit position kept as Qp = C - P the word pair of a 32-bit window is (Qp >> 5, Qp >> 5 + 1) and
//             v_alignbit(hi, lo, Qp) is the window, also on a word boundary.  13 words per lane: an odd stride keeps the
//             NT lanes on different LDS banks while they read the same word of their segments.
//   phase 1   lane j runs up through the kBlkGuessBits in front of its segment from an ASSUMED code boundary, notes
//             the first code that starts inside its segment (f_j), then counts codes and sums residuals up to the
//             first code that starts behind it (e_j).  If lane j-1 was in step, e_{j-1} == f_j; lane 0 of block 0
//             starts exactly, so the chain of equalities proves every lane exact.  A lane whose start does not
//             match its predecessor's end restarts from that end until nothing changes (once in a blue moon for
//             noise; a slope-1 ramp, whose codes all have the same length, never falls into step and takes up to NT
//             rounds -- correct, just slow).
//   blocks    lane 0 of block b > 0 checks its f_0 against the end block b-1 published after ITS phase 1 (one hop, no
//             chain); samples and the running sum in front of a block come from a decoupled look-back over the
//             blocks of the waveform ({status | count | sum} entries, tickets as in the encoder).  A block that has to
//             correct its start after it published its end flags the waveform; flagged waveforms are decoded again
//             by the one-workgroup-per-waveform kernel (drx_kernels.hip), which is also the one that judges them.
//   phase 2   every lane decodes its cnt_j codes again from f_j, now with the running sum, into an LDS staging
//             buffer in OUTPUT order (its first sample index is the prefix sum of the counts); the block then
//             copies whole aligned 128-byte lines to HBM.  (Scattered 2- and 8-byte stores of the previous
//             long-waveform kernel cost 2.5-5.4 x the output size in HBM writes: profiles/r02_*_pmc_traffic.json.)
// Two parses of every bit (1.4 + 1), no iteration in the common case, no sample ever stored twice.
// Delta filter only: the prefix sum over residual sums is what makes blocks independent.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "drx_device.h"
#include "drx_internal.h"

namespace drx {

constexpr int kBlkSegW = 13;             // words per lane (odd)
constexpr uint32_t kBlkPre = 8;          // words kept in front of a block: lane 0's run-up
constexpr uint32_t kBlkGuessBits = 160;  // run-up in front of a segment (a parse is in step after a few codes; 96, 128
                                         // and 224 bits measured: within 4 % of one another, profiles/r02_notes.md)
constexpr uint32_t kBlkTail = 4;         // words behind a block: a code that starts inside may end 24 bits behind it,
                                         // and a window reads three words

template <int NT>
struct BlkGeom {
    static constexpr uint32_t kWords = NT * kBlkSegW;                       // payload words per block
    static constexpr uint32_t kLdsWords = kBlkPre + kWords + kBlkTail + 4;  // + up to 3 words of 16-byte alignment
    static constexpr uint32_t kOutCap = NT * 72;                            // samples staged per copy-out (a segment holds
                                                                            // 64 at 6.5 bits per sample; more: further passes)
    static_assert(kLdsWords % 4 == 0, "the image is filled by 16-byte pieces");
};

__host__ __device__ inline uint32_t blk_words(uint32_t nt) { return nt * kBlkSegW; }

// unit = (waveform, block); unit_first[g] = first unit of waveform g, unit_first[W] = their number.  One workgroup.
__global__ __launch_bounds__(1024) void k_blk_units(uint64_t total_waves, const uint32_t *__restrict__ wave_words,
                                                    uint32_t words_per_block, uint32_t *__restrict__ unit_first) {
    __shared__ uint32_t wsum[16];
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    uint32_t run = 0;
    for (uint64_t i0 = 0; i0 < total_waves; i0 += 1024) {
        const uint64_t i = i0 + threadIdx.x;
        const uint32_t v = (i < total_waves) ? (wave_words[i] + words_per_block - 1u) / words_per_block : 0u;
        const uint32_t inc = wave_incl_scan_dpp(v);
        if (lane == 63) wsum[wv] = inc;
        __syncthreads();
        uint32_t before = 0, all = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) { before += (w < wv) ? wsum[w] : 0u; all += wsum[w]; }
        if (i < total_waves) unit_first[i] = run + before + inc - v;
        run += all;
        __syncthreads();
    }
    if (threadIdx.x == 0) unit_first[total_waves] = run;
}

enum { kBlkSkip = 0, kBlkCount = 1, kBlkValue = 2 };

// count-leading-zeros that is defined for 0 (any value will do there: an all-zero window exists only in a corrupt stream)
__device__ __forceinline__ uint32_t clz_nz(uint32_t x) { return (uint32_t)__builtin_clz(x | 1u); }

// Two codes from the 64-bit window at Qp (three words: a 64-bit window always holds two codes of at most 25 bits).
// W: the block's LDS image; word w of the image sits at W[K + 1 - w] and Qp = 32 K - (bit position), so that
// W[Qp >> 5 .. + 2] are the window's words, last one first, and v_alignbit(hi, lo, Qp) is its first half -- also on a
// word boundary (as in k_decode_lanes).  nu = minus the code length; v_bfe_u32 / v_alignbit_b32 read 5 bits of their
// offset / shift, ~t == 31 - t (mod 32) serves both.
struct BlkPair { uint32_t nu1, nu2, z1, z2; };
template <bool VALUES>
__device__ __forceinline__ BlkPair blk_pair(const uint32_t *W, uint32_t k, uint32_t Qp) {
    const uint32_t idx = Qp >> 5;
    const uint32_t lo2 = W[idx], lo = W[idx + 1u], hi = W[idx + 2u];
    const uint32_t winA = __builtin_amdgcn_alignbit(hi, lo, Qp);
    const uint32_t winB = __builtin_amdgcn_alignbit(lo, lo2, Qp);
    const uint32_t q1 = clz_nz(winA);
    const uint32_t kk1 = (winA < (1u << 24)) ? 16u : k;  // escape: eight zeros (:223-228)
    BlkPair r;
    r.nu1 = ~(q1 + kk1);
    const uint32_t win2 = __builtin_amdgcn_alignbit(winA, winB, r.nu1);
    const uint32_t q2 = clz_nz(win2);
    const uint32_t kk2 = (win2 < (1u << 24)) ? 16u : k;
    r.nu2 = ~(q2 + kk2);
    r.z1 = r.z2 = 0;
    if (VALUES) {
        r.z1 = (q1 << kk1) + __builtin_amdgcn_ubfe(winA, r.nu1, kk1);
        r.z2 = (q2 << kk2) + __builtin_amdgcn_ubfe(win2, r.nu2, kk2);
    }
    return r;
}
__device__ __forceinline__ uint32_t unzigzag(uint32_t z) { return (z >> 1) ^ (0u - (z & 1u)); }  // :172-177

// The parse.
//   kBlkSkip / kBlkCount: codes are taken while they START before the limit (Qp > qlim);
//   kBlkValue: exactly `cmax` codes (c counts them), each running sum (:80-89) stored as int16 at outp[c].
// A lane that is not enabled keeps its state.  (A variant that ran a wave without per-code masks while every lane had
// room for two more codes, and only the last few codes masked, was measured 4-8 % SLOWER: the vote per pair and the
// second loop cost what the masks had: profiles/r02_notes.md.)
template <int MODE>
__device__ __forceinline__ void blk_parse(const uint32_t *W, uint32_t k, bool enable, uint32_t &Qp, uint32_t qlim,
                                          uint32_t &c, uint32_t &sum, uint32_t cmax, uint16_t *outp) {
    auto more = [&](uint32_t q, uint32_t cc) __attribute__((always_inline)) {
        return enable && (MODE == kBlkValue ? cc < cmax : (int32_t)(q - qlim) > 0);
    };
    while (__any(more(Qp, c))) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {  // one vote per four codes
            const BlkPair p = blk_pair<MODE != kBlkSkip>(W, k, Qp);
            const bool act1 = more(Qp, c);
            const uint32_t Qa = Qp + p.nu1;
            const bool act2 = act1 && more(Qa, c + 1u);
            if (MODE != kBlkSkip) {
                const uint32_t s1 = sum + unzigzag(p.z1);
                const uint32_t s2 = s1 + unzigzag(p.z2);
                if (MODE == kBlkValue) {
                    if (act1) outp[c] = (uint16_t)s1;
                    if (act2) outp[c + 1u] = (uint16_t)s2;
                }
                sum = act2 ? s2 : (act1 ? s1 : sum);
            }
            c += (act1 ? 1u : 0u) + (act2 ? 1u : 0u);
            Qp = act2 ? Qa + p.nu2 : (act1 ? Qa : Qp);
        }
    }
}

template <int NT>
__global__ __launch_bounds__(NT) void k_decode_blocks(Geom G, const uint32_t *__restrict__ in, uint64_t in_words,
                                                      const uint64_t *__restrict__ wave_off,
                                                      const uint32_t *__restrict__ wave_words,
                                                      const uint32_t *__restrict__ unit_first, uint64_t *__restrict__ state,
                                                      uint32_t *__restrict__ ends, uint32_t *__restrict__ ticket,
                                                      uint32_t *__restrict__ fail, uint32_t *__restrict__ suspect,
                                                      DevStatus *st, int16_t *__restrict__ out) {
    using BG = BlkGeom<NT>;
    constexpr uint32_t K = BG::kLdsWords + 2u;  // word w of the image sits at W[K + 1 - w]; K = 2 (mod 4): 16-byte quads
    constexpr uint32_t C = 32u * K;
    constexpr int NW = NT / 64;
    constexpr uint32_t kSegBits = 32u * kBlkSegW;
    // one LDS object, the image first: its three-word windows are read with immediate offsets from address 0
    constexpr uint32_t kWSize = BG::kLdsWords + 4u, kObufWords = (BG::kOutCap + 16u) / 2u;
    __shared__ __attribute__((aligned(16))) uint32_t lds[kWSize + kObufWords + NT + 2 * NW + 4 + 2];
    uint32_t *const W = lds;
    uint16_t *const obuf = reinterpret_cast<uint16_t *>(lds + kWSize);
    uint32_t *const s_e = lds + kWSize + kObufWords;
    uint32_t(*const s_tot)[NW] = reinterpret_cast<uint32_t(*)[NW]>(s_e + NT);
    uint64_t *const s_b = reinterpret_cast<uint64_t *>(s_e + NT + 2 * NW);  // (kWSize, kObufWords, NT, 2 NW: all even)
    uint32_t &s_unit = s_e[NT + 2 * NW + 4], &s_pred = s_e[NT + 2 * NW + 5];
    const uint32_t tid = threadIdx.x;
    const int lane = lane_id(), wv = (int)(tid >> 6);
    const uint32_t k = G.k;
    const uint32_t total_units = unit_first[G.total_waves];
    const bool vec_ok = ((uintptr_t)in & 15u) == 0;
    typedef uint16_t __attribute__((address_space(1))) g_u16;
    typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
    typedef u32x4v __attribute__((address_space(1))) g_uint4;

    for (;;) {
        // units by ticket: every lower unit is held by a running (or finished) workgroup, so waiting for a
        // predecessor cannot deadlock whatever the dispatch order; the grid is sized to be resident
        if (tid == 0) s_unit = atomicAdd(ticket, 1u);
        __syncthreads();
        const uint32_t unit = s_unit;
        __syncthreads();
        if (unit >= total_units) return;
        uint32_t glo = 0, ghi = (uint32_t)G.total_waves;  // last g with unit_first[g] <= unit
        while (ghi - glo > 1u) {
            const uint32_t mid = (glo + ghi) >> 1;
            if (unit_first[mid] <= unit) glo = mid; else ghi = mid;
        }
        const uint64_t g = glo;
        const uint32_t blk = unit - unit_first[g];
        const WaveRef r = locate(G, g);
        const uint64_t pay_lo = wave_off[g] + 1u;
        const uint32_t n = wave_words[g], len = r.len;
        int16_t *y = out + r.sample_off;
        const uint32_t n_blocks = (n + BG::kWords - 1u) / BG::kWords;
        const uint32_t w0 = blk * BG::kWords;
        const uint32_t avail = (n - w0 < BG::kWords) ? n - w0 : BG::kWords;

        // ---- the block's image in LDS: words [w0 - kBlkPre, w0 + kWords + kBlkTail), from a 16-byte boundary ----
        const int64_t a_first = (int64_t)(pay_lo + w0) - (int64_t)kBlkPre;
        const int64_t al = a_first & ~(int64_t)3;
        const uint32_t s_i0 = (uint32_t)((int64_t)(pay_lo + w0) - al);  // image index of the block's first word
        const int64_t pay_hi = (int64_t)(pay_lo + n);                    // nothing behind the payload is read as stream
        for (uint32_t q = tid; q < BG::kLdsWords / 4u; q += NT) {
            const int64_t a = al + 4 * (int64_t)q;
            uint4 v;
            if (vec_ok && a >= 0 && a + 4 <= (int64_t)in_words && a + 4 <= pay_hi) {
                v = *reinterpret_cast<const uint4 *>(in + a);
            } else {
                auto ld = [&](int64_t i) { return (i >= 0 && i < (int64_t)in_words && i < pay_hi) ? in[i] : 0u; };
                v = make_uint4(ld(a), ld(a + 1), ld(a + 2), ld(a + 3));
            }
            // words 4q .. 4q+3 at W[K - 4q - 2 .. K - 4q + 1]: one 16-byte store (K - 4q - 2 = kLdsWords - 4q)
            *reinterpret_cast<uint4 *>(W + (BG::kLdsWords - 4u * q)) = make_uint4(v.w, v.z, v.y, v.x);
        }
        __syncthreads();

        // ---- phase 1: where the codes of my segment start, how many there are, what they sum to ----
        const uint32_t B0 = 32u * s_i0, bend = B0 + 32u * avail;
        const uint32_t bj = B0 + kSegBits * tid;
        const bool active = bj < bend;
        const uint32_t lim = (bj + kSegBits < bend) ? bj + kSegBits : bend;
        const bool exact0 = tid == 0 && blk == 0;  // the waveform's first code starts at bit 0
        uint32_t Qp = C - ((exact0 || !active) ? B0 : bj - kBlkGuessBits);
        uint32_t cnt = 0, sum = 0, dummy_c = 0, dummy_s = 0;
        blk_parse<kBlkSkip>(W, k, active && !exact0, Qp, C - bj, dummy_c, dummy_s, 0u, nullptr);
        if (exact0 || !active) Qp = C - B0;
        uint32_t f = C - Qp;  // first code that starts in my segment
        blk_parse<kBlkCount>(W, k, active, Qp, C - lim, cnt, sum, 0u, nullptr);
        if (!active) { cnt = 0; sum = 0; }
        uint32_t e = C - Qp;  // first code that starts behind it

        // every lane must start where its predecessor ended; lanes that do not, start again from there
        auto settle = [&]() __attribute__((always_inline)) {
            for (uint32_t it = 0; it <= (uint32_t)NT; ++it) {
                s_e[tid] = e;
                __syncthreads();
                const uint32_t want = tid ? s_e[tid - 1u] : f;
                const bool changed = active && want != f;
                if (!__syncthreads_or(changed ? 1 : 0)) break;  // (also: every read of s_e is done before the next write)
                if (changed) { f = want; Qp = C - f; cnt = 0; sum = 0; }
                blk_parse<kBlkCount>(W, k, changed, Qp, C - lim, cnt, sum, 0u, nullptr);
                if (changed) e = C - Qp;
            }
        };
        settle();
        const uint32_t last_active = (avail + (uint32_t)kBlkSegW - 1u) / (uint32_t)kBlkSegW - 1u;
        const uint32_t e_last0 = s_e[last_active];
        // the end of this block = the start of the next one, published as soon as it is known
        if (tid == 0) __hip_atomic_store(ends + unit, 0x80000000u | (e_last0 - bend), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (blk > 0) {
            if (tid == 0) {
                uint32_t v = 0, spins = 0;
                for (;;) {
                    v = __hip_atomic_load(ends + unit - 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (v & 0x80000000u) break;
                    __builtin_amdgcn_s_sleep(2);
                    if (++spins > (1u << 24)) { atomicOr(&st->err, kErrInternal); break; }  // cannot happen; never hang
                }
                s_pred = v & 0xffffu;
            }
            __syncthreads();
            const uint32_t true_f0 = B0 + s_pred;
            const bool fix0 = tid == 0 && true_f0 != f;
            if (__syncthreads_or(fix0 ? 1 : 0)) {
                if (fix0) { f = true_f0; Qp = C - f; cnt = 0; sum = 0; }
                blk_parse<kBlkCount>(W, k, fix0, Qp, C - lim, cnt, sum, 0u, nullptr);
                if (fix0) e = C - Qp;
                settle();
                // my successor has started from the end I published: if that end moved, its block is wrong
                if (tid == 0 && s_e[last_active] != e_last0 && blk + 1u < n_blocks) atomicExch(fail + g, 1u);
            }
        }

        // ---- samples and residual sum in front of my segment (workgroup scan) and in front of the block (look-back) ----
        const uint32_t incl_c = wave_incl_scan_dpp(cnt), incl_s = wave_incl_scan_dpp(sum);
        if (lane == 63) { s_tot[0][wv] = incl_c; s_tot[1][wv] = incl_s; }
        __syncthreads();
        uint32_t pre_c = 0, pre_s = 0, tot_c = 0, tot_s = 0;
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const uint32_t tc = s_tot[0][i], ts = s_tot[1][i];
            if (i < wv) { pre_c += tc; pre_s += ts; }
            tot_c += tc;
            tot_s += ts;
        }
        if (wv == 0) {
            const uint64_t mine = ((uint64_t)tot_c << 16) | (uint64_t)(tot_s & 0xffffu);
            uint64_t ex_c = 0, ex_s = 0;
            if (blk == 0) {
                if (lane == 0) __hip_atomic_store(state + unit, kScanPrefix | mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                if (lane == 0) __hip_atomic_store(state + unit, kScanAgg | mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int64_t first = (int64_t)unit - (int64_t)blk;  // block 0 of this waveform
                int64_t base = (int64_t)unit - 1;
                uint32_t spins = 0;
                for (;;) {
                    const int64_t i0 = base - lane;
                    uint64_t sv = kScanPrefix;  // in front of block 0: an empty prefix
                    if (i0 >= first) sv = __hip_atomic_load(state + i0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const uint32_t stt = (uint32_t)(sv >> 62);
                    const uint64_t pm = __ballot(stt == 2u), zm = __ballot(stt == 0u);
                    const int fp = pm ? __builtin_ctzll(pm) : 64;
                    const uint64_t nearer = fp >= 64 ? ~0ull : ((1ull << fp) - 1ull);
                    if (zm & nearer) {
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > (1u << 22)) { if (lane == 0) atomicOr(&st->err, kErrInternal); break; }
                        continue;
                    }
                    const uint64_t val = (lane <= fp) ? (sv & kScanValMask) : 0ull;
                    ex_c += wave_sum_u64(val >> 16);
                    ex_s += wave_sum_u64(val & 0xffffull);
                    if (fp < 64) break;
                    base -= 64;
                }
                if (lane == 0)
                    __hip_atomic_store(state + unit, kScanPrefix | ((((ex_c + tot_c) << 16) | ((ex_s + tot_s) & 0xffffull)) & kScanValMask),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (lane == 0) { s_b[0] = ex_c; s_b[1] = ex_s; }
        }
        __syncthreads();
        const uint64_t base_c = s_b[0];
        const uint32_t acc_base = (uint32_t)s_b[1];
        // the waveform has `len` samples; a code decoded out of the zero padding behind the last one does not count
        const uint32_t blk_first = base_c < (uint64_t)len ? (uint32_t)base_c : len;
        const uint32_t blk_count = (tot_c < len - blk_first) ? tot_c : len - blk_first;
        const uint32_t rel0 = pre_c + incl_c - cnt;  // my first sample, relative to the block's first
        const uint32_t todo = rel0 >= blk_count ? 0u : ((cnt < blk_count - rel0) ? cnt : blk_count - rel0);
        // the stream ended before the waveform did: a verdict for the kernel that runs after this one (a block behind a
        // mis-started one counts garbage, and its waveform is flagged for the fallback anyway)
        if (tid == 0 && blk + 1u == n_blocks && base_c + tot_c < (uint64_t)len) atomicExch(suspect + g, 1u);

        // ---- phase 2: the samples, staged in output order, whole lines to HBM ----
        uint32_t c = 0, acc = acc_base + pre_s + incl_s - sum;
        Qp = C - (todo ? f : B0);
        const uint32_t a0 = (uint32_t)((((uintptr_t)(y + blk_first)) >> 1) & 7u);  // kOutCap is a multiple of 8: the same every pass
        for (uint32_t R0 = 0; R0 < blk_count; R0 += BG::kOutCap) {
            // my samples with block-relative index below R0 + kOutCap
            const uint32_t cmax = (rel0 >= R0 + BG::kOutCap) ? 0u : ((todo < R0 + BG::kOutCap - rel0) ? todo : R0 + BG::kOutCap - rel0);
            // slot of sample c: a0 + rel0 + c - R0 (>= a0 for every c this pass decodes)
            uint16_t *outp = obuf + (int32_t)(a0 + rel0 - R0);
            blk_parse<kBlkValue>(W, k, c < cmax, Qp, 0u, c, acc, cmax, outp);
            __syncthreads();
            const uint32_t nsamp = (blk_count - R0 < BG::kOutCap) ? blk_count - R0 : BG::kOutCap;
            g_u16 *gbase = (g_u16 *)(y + blk_first + R0) - a0;  // 16-byte aligned
            const uint32_t np = (a0 + nsamp + 7u) >> 3;
            for (uint32_t p = tid; p < np; p += NT) {
                const uint32_t s_lo = 8u * p;
                if (s_lo >= a0 && s_lo + 8u <= a0 + nsamp) {
                    const uint4 v = *reinterpret_cast<const uint4 *>(obuf + s_lo);
                    *(g_uint4 *)(gbase + s_lo) = (u32x4v){v.x, v.y, v.z, v.w};
                } else {
#pragma unroll
                    for (uint32_t j = 0; j < 8u; ++j)
                        if (s_lo + j >= a0 && s_lo + j < a0 + nsamp) gbase[s_lo + j] = obuf[s_lo + j];
                }
            }
            __syncthreads();
        }
        // the waveform's last code must end in its last payload word: n_i = ceil(bits / 32) (src/deltaRice.c:237-241)
        if (todo && c == todo && (uint64_t)blk_first + rel0 + todo == (uint64_t)len) {
            const uint32_t bits_in_block = (C - Qp) - B0;
            if (w0 + ((bits_in_block + 31u) >> 5) != n) atomicExch(suspect + g, 1u);
        }
        __syncthreads();  // W, obuf and the s_* words are rewritten by the next unit
    }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
// Which batches take this decoder: uniform, delta filter, and fewer waveforms than a quarter of the lanes of the chip.
// Measured crossover against k_decode_lanes (tools/len_sweep.py, profiles/r02_blocks_vs_lanes.txt): a lane decodes
// ~17 samples per microsecond whatever else runs, so the lane kernel takes ~60 ns x WaveformLength until the chip is
// full (98 304 lanes); this kernel decodes ~400 samples per nanosecond once it has a few hundred blocks: the two
// meet near 24 000 waveforms for every WaveformLength from 4096 to 65 536.
bool blocks_batch(const Geom &G) {
    return G.uniform && G.n_taps == 0 && G.total_waves <= 24576u && G.u_wave_len >= 2048u;
}

static int blocks_nt(const Geom &G) {
    // lanes per block: a waveform of about (k + 3.5) bits per sample should fill most of its last block
    const uint64_t typ_words = ((uint64_t)G.u_wave_len * (2u * G.k + 7u)) >> 6;
    return typ_words <= blk_words(64) ? 64 : (typ_words <= blk_words(128) ? 128 : 256);
}

static uint64_t blocks_units_max(const Geom &G) {
    const uint64_t per = (max_payload_words(G.u_wave_len) + blk_words(blocks_nt(G)) - 1u) / blk_words(blocks_nt(G));
    return G.total_waves * (per ? per : 1u);
}

// scratch: u32 unit_first[W + 1] | u32 fail[W] | u32 suspect[W] | u32 ticket[1] (+ pad to 8 bytes) | u32 ends[units] | u64 state[units]
struct BlkScratch {
    uint32_t *unit_first, *fail, *suspect, *ticket, *ends;
    uint64_t *state;
    uint64_t bytes;
};
static BlkScratch blocks_layout(const Geom &G, void *base) {
    const uint64_t W = G.total_waves, U = blocks_units_max(G);
    uint64_t n32 = (W + 1u) + W + W + 1u;
    n32 = (n32 + 3u) & ~3ull;
    BlkScratch L;
    uint32_t *p = reinterpret_cast<uint32_t *>(base);
    L.unit_first = p;
    L.fail = p + (W + 1u);
    L.suspect = L.fail + W;
    L.ticket = L.suspect + W;
    L.ends = p + n32;
    const uint64_t ends32 = (U + 1u) & ~1ull;
    L.state = reinterpret_cast<uint64_t *>(L.ends + ends32);
    L.bytes = (n32 + ends32) * 4u + U * 8u;
    return L;
}

uint64_t blocks_scratch_bytes(const Geom &G) { return blocks_batch(G) ? blocks_layout(G, nullptr).bytes : 0; }

hipError_t launch_decode_blocks(const Geom &G, const uint32_t *d_in, uint64_t in_words, const uint64_t *d_wave_off,
                                const uint32_t *d_wave_words, void *d_blk, DevStatus *d_status, int16_t *d_out,
                                const uint32_t **fail_out, const uint32_t **suspect_out, hipStream_t s) {
    const BlkScratch L = blocks_layout(G, d_blk);
    hipError_t e = hipMemsetAsync(d_blk, 0, L.bytes, s);
    if (e != hipSuccess) return e;
    const int nt = blocks_nt(G);
    k_blk_units<<<1, 1024, 0, s>>>(G.total_waves, d_wave_words, blk_words((uint32_t)nt), L.unit_first);
    // resident grid: 256 CUs x workgroups per CU (LDS: 51 KB at NT = 256, 26 KB at 128, 13 KB at 64), never more than there are units
    const uint64_t units = blocks_units_max(G);
    if (nt == 64) {
        const unsigned grid = (unsigned)(units < 256u * 12u ? units : 256u * 12u);
        k_decode_blocks<64><<<grid, 64, 0, s>>>(G, d_in, in_words, d_wave_off, d_wave_words, L.unit_first, L.state, L.ends, L.ticket,
                                               L.fail, L.suspect, d_status, d_out);
    } else if (nt == 128) {
        const unsigned grid = (unsigned)(units < 256u * 6u ? units : 256u * 6u);
        k_decode_blocks<128><<<grid, 128, 0, s>>>(G, d_in, in_words, d_wave_off, d_wave_words, L.unit_first, L.state, L.ends, L.ticket,
                                                 L.fail, L.suspect, d_status, d_out);
    } else {
        const unsigned grid = (unsigned)(units < 256u * 3u ? units : 256u * 3u);
        k_decode_blocks<256><<<grid, 256, 0, s>>>(G, d_in, in_words, d_wave_off, d_wave_words, L.unit_first, L.state, L.ends, L.ticket,
                                                 L.fail, L.suspect, d_status, d_out);
    }
    *fail_out = L.fail;
    *suspect_out = L.suspect;
    return hipGetLastError();
}

}  // namespace drx
