// drx_blocks.hip -- block-parallel decoder: a WORKGROUP per block of a waveform's stream.
//
// The lane-per-waveform decoder (k_decode_lanes, drx_decode_kernels.hip) needs ~10^5 waveforms to fill an MI355X.  The
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
//             by the one-workgroup-per-waveform kernel (drx_decode_kernels.hip), which is also the one that judges them.
//   phase 2   phase 1 has left every lane's running sums (relative to its first code) in LDS, lane-major; once the
//             prefix sums over counts and sums are known each lane reads its own into registers, adds its base and
//             writes them to the same buffer in OUTPUT order; the block then copies whole aligned 128-byte lines to
//             HBM.  (Scattered 2- and 8-byte stores of the previous long-waveform kernel cost 2.5-5.4 x the output
//             size in HBM writes: profiles/r02_*_pmc_traffic.json.)  A block in which some lane holds more codes than
//             its share of the buffer (kBlkLaneCap; long runs of tiny residuals) decodes a second time instead, in
//             as many staging passes as it needs.
// 1.4 parses of every bit in the common case, no iteration, no sample ever stored twice.
// Delta filter only: the prefix sum over residual sums is what makes blocks independent.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <type_traits>

#include "drx_device.h"
#include "drx_internal.h"
#include "drx_iir_math.h"

namespace drx {

// Words per lane (odd: the lanes' windows fall on different banks) and the samples a lane can leave in its share of the
// staging buffer.  A block is a fixed number of BITS and costs about the same whatever it holds, so one geometry for all
// data loses both ways: with 352 bits per lane noisy data (12.75 bits per sample: 27 samples per lane) decoded at 0.18
// instead of 0.25 of the roofline, and quiet data under the reference's default RiceParameter (4 bits per sample: 88
// samples per lane, more than the 76 its share holds, so EVERY block took the second parse) at 0.16.  The class is chosen
// per decode from what the caller states about the stream -- 32 in_words / total_samples bits per sample, so a
// RiceParameter that does not suit the data is covered too -- for ~50 samples per lane with 15 % to spare, and every class
// takes 48-50 dwords of LDS per lane:
//   words per lane        9      11      15      19
//   samples staged       76      76      68      60
//   bits per sample   < 5.4   < 8.2  < 11.7    above
// Plan-time estimates (which decoder a batch takes, lanes per block) assume the class that suits the RiceParameter;
// scratch is sized for the smallest blocks.
constexpr int kBlkSegWMin = 9;
__host__ __device__ constexpr int blk_segw_plan(uint32_t k) { return k <= 4u ? 11 : (k <= 7u ? 15 : 19); }
__host__ __device__ constexpr int blk_segw_bits10(uint64_t b10) { return b10 >= 117u ? 19 : (b10 >= 82u ? 15 : (b10 >= 54u ? 11 : 9)); }
__host__ __device__ constexpr uint32_t blk_lane_cap(int segw) { return segw <= 11 ? 76u : (segw == 15 ? 68u : 60u); }
constexpr uint32_t kBlkPre = 8;          // words kept in front of a block: lane 0's run-up
#ifndef DRX_BLK_GUESS_BITS
#define DRX_BLK_GUESS_BITS 128
#endif
constexpr uint32_t kBlkGuessBits = DRX_BLK_GUESS_BITS;  // run-up in front of a segment (a parse is in step after a few codes; 96, 128,
                                         // 160 and 224 bits measured: within 4 % of one another, profiles/r02_notes.md)
#ifndef DRX_BLK_ROUNDS
#define DRX_BLK_ROUNDS 8
#endif
constexpr uint32_t kBlkRounds = DRX_BLK_ROUNDS;  // tickets per resident workgroup a launch should at least have (see run_len)
constexpr uint32_t kBlkTail = 4;         // words behind a block: a code that starts inside may end 24 bits behind it,
                                         // and a window reads three words

template <int NT, int SW>
struct BlkGeom {
    static constexpr int kSegW = SW;
    static constexpr uint32_t kLaneCap = blk_lane_cap(SW);                  // samples a lane can stage
    static constexpr uint32_t kLaneStride = kLaneCap / 2u + 1u;             // dwords per lane: the samples + a dump slot (odd: no bank conflicts)
    static constexpr uint32_t kWords = NT * SW;                             // payload words per block
    static constexpr uint32_t kLdsWords = kBlkPre + kWords + kBlkTail + 4;  // + up to 3 words of 16-byte alignment
    static constexpr uint32_t kOutCap = NT * kLaneCap;                      // samples staged per copy-out
    static constexpr uint32_t kStageWords = NT * kLaneStride;               // the staging buffer, lane-major or output order
    static_assert(kLdsWords % 4 == 0, "the image is filled by 16-byte pieces");
};

__host__ __device__ inline uint32_t blk_words(uint32_t nt, uint32_t k) { return nt * (uint32_t)blk_segw_plan(k); }  // plan-time estimates
__host__ __device__ inline uint32_t blk_words_min(uint32_t nt) { return nt * (uint32_t)kBlkSegWMin; }               // scratch sizing

// Most blocks any waveform of the batch has: info[0]; tickets of the decode launch: info[1] = info[0] x waveforms.
// One workgroup.
// (list: the waveforms of this launch, nullptr = all of them in order)
__global__ __launch_bounds__(1024) void k_blk_max(uint64_t total_waves, const uint32_t *__restrict__ wave_words,
                                                  uint32_t words_per_block, uint32_t *__restrict__ info,
                                                  const uint32_t *__restrict__ list) {
    __shared__ uint32_t wmax[16];
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    uint32_t m = 0;
    for (uint64_t i = threadIdx.x; i < total_waves; i += 1024) {
        const uint32_t v = (wave_words[list ? list[i] : i] + words_per_block - 1u) / words_per_block;
        m = v > m ? v : m;
    }
    m = wave_max_u32(m);
    if (lane == 0) wmax[wv] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w) m = wmax[w] > m ? wmax[w] : m;
        info[0] = m;
        const uint64_t t = (uint64_t)m * total_waves;
        info[1] = t > 0xffffffffull ? 0xffffffffu : (uint32_t)t;
    }
}

// Workgroup barrier for data exchanged through LDS only.  __syncthreads() also fences global memory, i.e. waits for
// every global load and store the wave has in flight (s_waitcnt vmcnt(0)): the image and the ticket fetched ahead and
// the output lines being written would all be waited for at the next barrier, which is exactly what fetching ahead is
// meant to avoid.  Nothing this kernel exchanges between the waves of a workgroup goes through global memory.
__device__ __forceinline__ void blk_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

enum { kBlkSkip = 0, kBlkCount = 1, kBlkValue = 2 };

// Diagnostic build (-DDRX_BLK_STAMPS, never shipped): thread 0 of every workgroup adds the cycles between phase
// boundaries into prof[]; launch_decode_blocks prints the shares.
#ifdef DRX_BLK_STAMPS
#define BLK_STAMP(i)                                                                                         \
    do {                                                                                                     \
        if (tid == 0) {                                                                                      \
            const unsigned long long t_now = __builtin_amdgcn_s_memtime();                                   \
            s_prof[(i)] += t_now - t_prev;                                                                   \
            t_prev = t_now;                                                                                  \
        }                                                                                                    \
    } while (0)
#else
#define BLK_STAMP(i) do { } while (0)
#endif

// count-leading-zeros that is defined for 0: v_ffbh_u32 returns -1 there, which with the escape's 16 payload bits moves the
// parse on by 15 + 1 bits -- any progress will do (an all-zero window is the padding behind a waveform or a corrupt stream)
__device__ __forceinline__ uint32_t clz_nz(uint32_t x) { return ffbh(x); }
// the 32 bits at Qp of a block's image (see blk_pair)
__device__ __forceinline__ uint32_t blk_window(const uint32_t *W, uint32_t Qp) {
    const uint32_t idx = Qp >> 5;
    return __builtin_amdgcn_alignbit(W[idx + 2u], W[idx + 1u], Qp);
}

// Two codes from the 64-bit window at Qp (three words: a 64-bit window always holds two codes of at most 25 bits).
// W: the block's LDS image; word w of the image sits at W[K + 1 - w] and Qp = 32 K - (bit position), so that
// W[Qp >> 5 .. + 2] are the window's words, last one first, and v_alignbit(hi, lo, Qp) is its first half -- also on a
// word boundary (as in k_decode_lanes).  nu = minus the code length; v_bfe_u32 / v_alignbit_b32 read 5 bits of their
// offset / shift, ~t == 31 - t (mod 32) serves both.
struct BlkPair { uint32_t nu1, nu2, z1, z2; bool pad1, pad2; };
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
    r.pad1 = winA < (1u << 23);  // nine zero bits: not a code
    r.pad2 = win2 < (1u << 23);
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
//     kBlkCount also leaves the running sums (:80-89, from 0 at the lane's first code) in `stage`, two per dword,
//     pair c / 2 at stage[min(c / 2, kBlkLaneCap / 2)] (the last dword is a dump slot), and stops at a window that
//     opens with nine zero bits: no code does (q < 8: '1' within nine bits, escape: eight zeros then '1',
//     :215-228), so that is the zero padding behind the waveform's last code, not a sample;
//   kBlkValue: exactly `cmax` codes (c counts them), each running sum stored as int16 at outp[c].
// A lane that is not enabled keeps its state.  (A variant that ran a wave without per-code masks while every lane had
// room for two more codes, and only the last few codes masked, was measured 4-8 % SLOWER: the vote per pair and the
// second loop cost what the masks had: profiles/r02_notes.md.)
// RESID: the RESIDUALS themselves are staged / stored instead of their running sums (general prediction filters: the inverse
// filter runs afterwards, in place, k_iir_tiles).
// qpad (kBlkCount): a window of nine zero bits is the waveform's padding only where padding can be -- inside its LAST payload
// word (Qp <= qpad; 0 = that word is not in this block).  Anywhere else it is what a parse that is not yet in step sees inside
// an escape's payload (z < 128 has nine leading zeros in its sixteen bits); a lane that stopped there reported a wrong END, its
// successor restarted from that end and stopped there too, and the correction crept through the block one lane per settle
// round: 256 rounds per block at m = 4 (25 % escapes), NOPTREX 26 ms instead of 2, 25 x 14 M samples 106 ms (round 3).
// PAD = false: the caller knows that qpad = 0 (every block but a waveform's last), and the padding test is compiled away.
template <int MODE, bool RESID = false, bool PAD = true>
__device__ __forceinline__ void blk_parse(const uint32_t *W, uint32_t k, bool enable, uint32_t &Qp, uint32_t qlim,
                                          uint32_t &c, uint32_t &sum, uint32_t cmax, uint16_t *outp, uint32_t *stage = nullptr,
                                          uint32_t qpad = 0u, uint32_t cap2 = 0u) {  // cap2: the dump slot = half the lane's share
    auto more = [&](uint32_t q, uint32_t cc) __attribute__((always_inline)) {
        return enable && (MODE == kBlkValue ? cc < cmax : (int32_t)(q - qlim) > 0);
    };
    // staging slot of the next pair (kBlkCount; c is even whenever a pair is staged): a dword index that saturates at the dump slot
    uint32_t slot = (c >> 1) < cap2 ? (c >> 1) : cap2;
    while (__builtin_amdgcn_ballot_w64(more(Qp, c)) != 0ull) {  // (__any() costs a v_cndmask and a v_cmp more)
#pragma unroll
        for (int u = 0; u < 2; ++u) {  // one vote per four codes
            const BlkPair p = blk_pair<MODE != kBlkSkip>(W, k, Qp);
            bool act1 = more(Qp, c);
            const uint32_t Qa = Qp + p.nu1;
            bool act2 = act1 && more(Qa, c + 1u);
            if (MODE == kBlkCount && PAD) {
                if (act1 && p.pad1 && (int32_t)(Qp - qpad) <= 0) { act1 = act2 = false; qlim = Qp; }  // (Qp stays: where the padding starts)
                if (act2 && p.pad2 && (int32_t)(Qa - qpad) <= 0) { act2 = false; qlim = Qa; }
            }
            if (MODE != kBlkSkip) {
                const uint32_t s1 = RESID ? unzigzag(p.z1) : sum + unzigzag(p.z1);
                const uint32_t s2 = RESID ? unzigzag(p.z2) : s1 + unzigzag(p.z2);
                if (MODE == kBlkValue) {
                    if (act1) outp[c] = (uint16_t)s1;
                    if (act2) outp[c + 1u] = (uint16_t)s2;
                }
                if (MODE == kBlkCount) {  // c is even here: only a lane's last pair can end after its first code
                    if (act1) stage[slot] = __builtin_amdgcn_perm(s2, s1, 0x05040100u);
                    slot = slot + 1u < cap2 ? slot + 1u : cap2;
                }
                sum = act2 ? s2 : (act1 ? s1 : sum);
            }
            c += (act1 ? 1u : 0u) + (act2 ? 1u : 0u);
            Qp = act2 ? Qa + p.nu2 : (act1 ? Qa : Qp);
        }
    }
}

// The count parse of a block that does not hold its waveform's last payload word (no padding to recognise): the limit is
// tested once per PAIR of codes and the pair's work runs under the lane's exec mask; a lane whose last pair's second code
// started at or behind the limit takes that code back after the loop.  (The general form above tests every code: a
// quarter more VALU instructions per sample.)  Qp stays far above zero in a block's image (C - bend >= 224 bits), so the
// limit test is an unsigned compare.
template <bool RESID>
__device__ __forceinline__ void blk_count_pairs(const uint32_t *W, uint32_t k, bool enable, uint32_t &Qp, uint32_t qlim,
                                                uint32_t &c, uint32_t &sum, uint32_t *stage, uint32_t cap2) {
    uint32_t slot = 0, Qa_l = Qp, s1_l = sum;  // (c = 0 on entry)
    // A lane that is not enabled gets a limit no position exceeds: the loop's condition is then ONE compare that is the exec
    // mask.  The loop is ROTATED (test at the bottom): with the test at the top the compiler kept the values used behind the
    // loop apart from the loop-carried ones and copied five registers there and back per trip (ten v_mov per four codes, a
    // seventh of the parse's VALU instructions).
    const uint32_t ql = enable ? qlim : 0xffffffffu;
    if (__builtin_amdgcn_ballot_w64(Qp > ql) != 0ull) {
        do {
#pragma unroll
            for (int u = 0; u < 2; ++u) {  // one vote per four codes
                if (Qp > ql) {
                    const BlkPair p = blk_pair<true>(W, k, Qp);
                    const uint32_t s1 = RESID ? unzigzag(p.z1) : sum + unzigzag(p.z1);
                    const uint32_t s2 = RESID ? unzigzag(p.z2) : s1 + unzigzag(p.z2);
                    stage[slot] = __builtin_amdgcn_perm(s2, s1, 0x05040100u);
                    Qa_l = Qp + p.nu1;
                    s1_l = s1;
                    sum = s2;
                    c += 2u;
                    Qp = Qa_l + p.nu2;
                }
                slot = slot + 1u < cap2 ? slot + 1u : cap2;
            }
        } while (__builtin_amdgcn_ballot_w64(Qp > ql) != 0ull);
    }
    if (enable && c != 0u && !(Qa_l > qlim)) { c -= 1u; Qp = Qa_l; sum = s1_l; }
}

// The run-up in the same form: codes are skipped, a pair at a time, while they start in front of the limit.
__device__ __forceinline__ void blk_skip_pairs(const uint32_t *W, uint32_t k, bool enable, uint32_t &Qp, uint32_t qlim) {
    uint32_t Qa_l = Qp;
    const uint32_t ql = enable ? qlim : 0xffffffffu;  // (as in blk_count_pairs: one compare, a rotated loop)
    if (__builtin_amdgcn_ballot_w64(Qp > ql) != 0ull) {
        do {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (Qp > ql) {
                    const BlkPair p = blk_pair<false>(W, k, Qp);
                    Qa_l = Qp + p.nu1;
                    Qp = Qa_l + p.nu2;
                }
            }
        } while (__builtin_amdgcn_ballot_w64(Qp > ql) != 0ull);
    }
    if (enable && !(Qa_l > qlim)) Qp = Qa_l;  // (no pair taken: Qa_l is Qp)
}

// FUSE (with RESID): the inverse of a general prediction filter (at most four taps, taps[0] = +-1; src/deltaRice.c:91-102) runs
// INSIDE this kernel, over a block's residuals while they sit in LDS in output order: samples, not residuals, are what reaches
// HBM, and no second pass (k_iir_tiles: 2 + 2 more bytes of traffic per sample) follows.  The recurrence is linear over
// Z / 2^16 (drx_iir.hip has the algebra): lane t owns samples [t M, (t + 1) M) of the staging buffer (M = its lane share, so
// equal run lengths and matrices from a table), pass 1 = zero-state response of every run (lane 0 starts from the state in
// front of the block instead), a scan over the lanes with A^(M 2^d), pass 2 = the runs again from their true states.  The state
// behind a block is its last three samples: it stays with thread 0 through a run of blocks, and goes from a run's last block
// to the next run's first through `xstate` (one 8-byte word per block slot, its own flag).  That hand-over is a serial chain
// along a waveform, so the host fuses only where runs of ONE waveform are rarely in flight together (as many waveforms as
// resident workgroups); elsewhere the two-pass form stays.
template <int NT, bool RESID, int SW, bool FUSE = false>
__global__ __launch_bounds__(NT) void k_decode_blocks(Geom G, const uint32_t *__restrict__ in, uint64_t in_words,
                                                      const uint64_t *__restrict__ wave_off,
                                                      const uint32_t *__restrict__ wave_words,
                                                      const uint32_t *__restrict__ info, uint32_t slots_per_wave,
                                                      uint32_t run_len, uint64_t *__restrict__ state,
                                                      uint32_t *__restrict__ ends, uint32_t *__restrict__ ticket,
                                                      uint32_t *__restrict__ fail, uint32_t *__restrict__ suspect,
                                                      DevStatus *st, int16_t *__restrict__ out, unsigned long long *prof,
                                                      const uint32_t *__restrict__ wave_list, uint32_t n_list,
                                                      const uint32_t *__restrict__ itab, uint64_t *__restrict__ xstate) {
    static_assert(!FUSE || RESID, "the fused inverse filter works on residuals");
    using BG = BlkGeom<NT, SW>;
    constexpr int kBlkSegW = SW;
    constexpr uint32_t kBlkLaneCap = BG::kLaneCap, kBlkLaneStride = BG::kLaneStride, kCap2 = BG::kLaneCap / 2u;
    constexpr uint32_t K = BG::kLdsWords + 2u;  // word w of the image sits at W[K + 1 - w]; K = 2 (mod 4): 16-byte quads
    constexpr uint32_t C = 32u * K;
    constexpr int NW = NT / 64;
    constexpr uint32_t kSegBits = 32u * kBlkSegW;
    // one LDS object, the image first: its three-word windows are read with immediate offsets from address 0
    constexpr uint32_t kWSize = BG::kLdsWords + 4u, kObufWords = BG::kStageWords;  // (kOutCap + 8 samples fit: NT >= 8)
    __shared__ __attribute__((aligned(16))) uint32_t lds[kWSize + kObufWords + NT + 2 * NW + 4 + 8 + 4 + 3 * NW];  // (+ s_unit, s_pred, s_next, s_vote[3], ..., FUSE: s_F[NW][3])
    uint32_t *const W = lds;
    uint32_t *const stage = lds + kWSize;                        // phase 1: lane-major, kBlkLaneStride dwords per lane
    uint16_t *const obuf = reinterpret_cast<uint16_t *>(stage);  // phase 2: the block's samples in output order
    uint32_t *const s_e = lds + kWSize + kObufWords;
    uint32_t(*const s_tot)[NW] = reinterpret_cast<uint32_t(*)[NW]>(s_e + NT);
    uint64_t *const s_b = reinterpret_cast<uint64_t *>(s_e + NT + 2 * NW);  // (kWSize, kObufWords, NT, 2 NW: all even)
    uint32_t &s_unit = s_e[NT + 2 * NW + 4], &s_pred = s_e[NT + 2 * NW + 5];
    uint32_t(*const s_F)[3] = reinterpret_cast<uint32_t(*)[3]>(s_e + NT + 2 * NW + 16);  // FUSE: a wavefront's zero-state response
    const uint32_t tid = threadIdx.x;
    const int lane = lane_id(), wv = (int)(tid >> 6);
    uint32_t k = G.k;
    asm volatile("" : "+v"(k));  // in a VGPR for good: the parse selects between k and 16 per code, and re-materialised it per pair
    // A ticket is a RUN of run_len consecutive blocks of one waveform, dealt run-major: run 0 of every waveform, then run 1
    // of every waveform, ...  Inside a run only its first block talks to other workgroups (where the predecessor's
    // stream ended, the look-back for samples and sum in front of it); the others start exactly where the block before
    // them ended, with counts carried in registers.  The host makes runs longer than one block only when there are more
    // waveforms than workgroups: two runs of ONE waveform in flight together serialise (the later one's look-back waits
    // for the earlier one's last block).  The image of the next block -- of this run or of the next ticket's -- travels
    // while the current block's samples are put in order and written out, and the next ticket is drawn a run ahead.
    // In-kernel stamps of the first version (a ticket per block, waveform-major: 768 blocks of ONE 14 M-sample waveform
    // in flight, every one polling twelve windows of aggregates) had 11 % of a workgroup's time in the parse and 77 % in
    // four waits: ticket, image, predecessor's end, look-back (profiles/r02_notes.md).
    // the waveforms of THIS launch: all of the batch, or (ragged batches) those of one length class -- tickets are dealt
    // run-major over them, and a class whose waveforms differ by less than 2x in length wastes few tickets on empty runs
    const uint32_t n_waves32 = wave_list ? n_list : (uint32_t)G.total_waves;
    const uint32_t max_runs = (info[0] + run_len - 1u) / run_len;
    const uint64_t total_units64 = (uint64_t)max_runs * n_waves32;
    const uint32_t total_units = total_units64 > 0xffffffffull ? 0xffffffffu : (uint32_t)total_units64;
    typedef uint16_t __attribute__((address_space(1))) g_u16;
    typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
    typedef u32x4v __attribute__((address_space(1))) g_uint4;
    constexpr int NQ = (int)((BG::kLdsWords / 4u + NT - 1u) / NT);  // 16-byte pieces of an image per thread

#ifdef DRX_BLK_STAMPS
    // (per workgroup in LDS, added to the launch's counters once at the end: an atomic per stamp on sixteen shared addresses was
    // itself what the stamped build waited for)
    __shared__ unsigned long long s_prof[16];
    if (tid < 16u) s_prof[tid] = 0ull;
    unsigned long long t_prev = __builtin_amdgcn_s_memtime();
    auto flush_prof = [&]() { if (tid == 0) for (int i = 0; i < 10; ++i) atomicAdd(prof + i, s_prof[i]); };
#else
    auto flush_prof = [] {};
#endif
    // tickets: every lower ticket is held by a running (or finished) workgroup, so waiting for a predecessor cannot
    // deadlock whatever the dispatch order; the grid is sized to be resident
    uint32_t &s_next = s_e[NT + 2 * NW + 6];
    uint32_t &s_front = s_e[NT + 2 * NW + 7];   // settle(): the lowest lane that started again last round and how far its end moved
    uint32_t &s_defer = s_e[NT + 2 * NW + 11];  // this block does not publish its end before its start is verified
    uint32_t *const s_vote = s_e + NT + 2 * NW + 8;  // [3], in rotation, so that a vote needs one barrier
    uint32_t vote_no = 0;
    // true in every thread if `v` holds in any thread of the workgroup.  Vote i uses word i mod 3; thread 0 clears the word
    // of vote i + 1 on its way into vote i: every thread has passed the barrier of vote i - 1 by then, so none still
    // reads the word of vote i - 2 (the same word), and the word of vote i - 1, which slow threads may still read, is another.
    auto wg_any = [&](bool v) __attribute__((always_inline)) {
        uint32_t *w = s_vote + vote_no % 3u;
        if (__any(v) && lane == 0) __hip_atomic_fetch_or(w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (tid == 0) s_vote[(vote_no + 1u) % 3u] = 0u;
        blk_barrier();
        ++vote_no;
        return *w != 0u;
    };
    if (tid == 0) { s_unit = atomicAdd(ticket, 1u); s_vote[0] = 0u; s_vote[1] = 0u; s_vote[2] = 0u; }
    __syncthreads();
    uint32_t unit = s_unit;
    __syncthreads();
    // where a ticket's run lies: waveform, first block, payload
    struct RunRef { uint64_t g, pay_lo; uint32_t n, blk_lo, blk_hi; };
    auto run_of = [&](uint32_t u) __attribute__((always_inline)) {
        RunRef q;
        const uint32_t run = u / n_waves32;
        const uint32_t j = u - run * n_waves32;
        q.g = wave_list ? wave_list[j] : j;
        q.pay_lo = wave_off[q.g] + 1u;
        q.n = wave_words[q.g];
        const uint32_t n_blocks = (q.n + BG::kWords - 1u) / BG::kWords;
        q.blk_lo = run * run_len;
        q.blk_hi = (q.blk_lo + run_len < n_blocks) ? q.blk_lo + run_len : n_blocks;  // (blk_lo >= n_blocks: an empty run)
        return q;
    };
    // image of block b: words [b kWords - kBlkPre, (b + 1) kWords + kBlkTail) of the payload, from a 16-byte boundary
    auto image_base = [&](uint64_t pay_lo, uint32_t b) { return ((int64_t)(pay_lo + (uint64_t)b * BG::kWords) - (int64_t)kBlkPre) & ~(int64_t)3; };
    // The fetch is unconditional 16-byte loads from a clamped address and nothing else: a load inside a branch is waited
    // for at the end of that branch, which would put the whole round trip back in front of the parse.  What the clamp
    // and the payload's end invalidate is sorted out when the registers are written to LDS.  (in_words >= 64 here: the
    // host gives this decoder waveforms of 2048 samples and more.)
    const int64_t a_max = (int64_t)in_words - 4;
    auto fetch_image = [&](const RunRef &q, uint32_t b, uint4 (&v)[NQ]) __attribute__((always_inline)) {
        const int64_t al = image_base(q.pay_lo, b);
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int64_t a = al + 4 * (int64_t)(tid + (uint32_t)i * NT);
            const int64_t ac = a < 0 ? 0 : (a > a_max ? a_max : a);
            v[i] = *reinterpret_cast<const uint4 *>(in + ac);
        }
    };
    auto store_image = [&](const RunRef &q, uint32_t b, const uint4 (&v)[NQ]) __attribute__((always_inline)) {
        const int64_t al = image_base(q.pay_lo, b);
        const int64_t pay_hi = (int64_t)(q.pay_lo + q.n);  // nothing behind the payload is read as stream
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const uint32_t qd = tid + (uint32_t)i * NT;
            const int64_t a = al + 4 * (int64_t)qd;
            if (qd >= BG::kLdsWords / 4u) continue;
            uint4 w = v[i];
            if (a < 0 || a > a_max) {  // the clamp moved this piece: word by word (the first and the last piece of a batch)
                auto ld = [&](int64_t j) { return (j >= 0 && j < (int64_t)in_words) ? in[j] : 0u; };
                w = make_uint4(ld(a), ld(a + 1), ld(a + 2), ld(a + 3));
            }
            w.x = (a + 0 < pay_hi) ? w.x : 0u;
            w.y = (a + 1 < pay_hi) ? w.y : 0u;
            w.z = (a + 2 < pay_hi) ? w.z : 0u;
            w.w = (a + 3 < pay_hi) ? w.w : 0u;
            // words 4q .. 4q+3 at W[K - 4q - 2 .. K - 4q + 1]: one 16-byte store (K - 4q - 2 = kLdsWords - 4q)
            *reinterpret_cast<uint4 *>(W + (BG::kLdsWords - 4u * qd)) = make_uint4(w.w, w.z, w.y, w.x);
        }
    };
    // ---- FUSE: the inverse filter over the samples [a0, a0 + nsamp) of the staging buffer, in place ----
    V3 xs{0u, 0u, 0u};  // the state in front of the next sample of the waveform (y[i-1], y[i-2], y[i-3]); thread 0's is the one that counts
    auto iir_lds = [&](uint32_t a0, uint32_t nsamp) __attribute__((always_inline)) {
        constexpr uint32_t M = kBlkLaneCap, NP = M / 2u;  // samples / dwords per lane
        static_assert(NP % 2u == 0u, "a lane's share is moved in 8-byte pieces");
        if (nsamp == 0u) return;
        const uint32_t c1 = itab[kRunTabC], c2 = itab[kRunTabC + 1], c3 = itab[kRunTabC + 2], sg = itab[kRunTabC + 3];
        uint32_t *const mine = reinterpret_cast<uint32_t *>(obuf) + tid * NP;
        // the recurrence over my run from state s, two dwords (four samples) of LDS at a time -- the run is read once per pass
        // rather than held in 38 registers across the scan (that form needed 214 registers: two workgroups per CU instead of
        // three).  Thread 0's first a0 entries lie in front of the block's first sample: no step there.  EMIT: the samples
        // replace the residuals.
        uint32_t last2 = 0, last1 = 0;  // EMIT: the run's last two dwords (the last lane may have to go on behind its share)
        auto run = [&](V3 s0, auto emit_tag) __attribute__((always_inline)) {
            constexpr bool EMIT = decltype(emit_tag)::value;
            uint32_t sx = s0.x, sy = s0.y, sz = s0.z;
            // one dword = two samples; SKIP: the dword may lie in front of the block's first sample (thread 0's first four)
            // (component by component: a select between two structs is compiled as a select between their ADDRESSES, and the
            // states went through scratch memory)
            auto pair = [&](uint32_t dj, uint32_t j, auto skip_tag) __attribute__((always_inline)) {
                constexpr bool SKIP = decltype(skip_tag)::value;
                const uint32_t lo = dj & 0xffffu, hi = dj >> 16;
                uint32_t a = __umul24(lo, sg) + __umul24(c1, sx) + __umul24(c2, sy) + __umul24(c3, sz);
                if (SKIP) {
                    const bool skip = tid == 0u && 2u * j < a0;
                    const uint32_t nx = skip ? sx : a, ny = skip ? sy : sx, nz = skip ? sz : sy;
                    a = skip ? lo : a;
                    sx = nx; sy = ny; sz = nz;
                } else {
                    sz = sy; sy = sx; sx = a;
                }
                uint32_t b = __umul24(hi, sg) + __umul24(c1, sx) + __umul24(c2, sy) + __umul24(c3, sz);
                if (SKIP) {
                    const bool skip = tid == 0u && 2u * j + 1u < a0;
                    const uint32_t nx = skip ? sx : b, ny = skip ? sy : sx, nz = skip ? sz : sy;
                    b = skip ? hi : b;
                    sx = nx; sy = ny; sz = nz;
                } else {
                    sz = sy; sy = sx; sx = b;
                }
                return __builtin_amdgcn_perm(b, a, 0x05040100u);
            };
            uint2 w = *reinterpret_cast<const uint2 *>(mine);
#pragma unroll
            for (uint32_t j0 = 0; j0 < 4u; j0 += 2u) {  // the first eight samples
                const uint2 wn = *reinterpret_cast<const uint2 *>(mine + j0 + 2u);
                const uint32_t o0 = pair(w.x, j0, std::true_type{}), o1 = pair(w.y, j0 + 1u, std::true_type{});
                if (EMIT) *reinterpret_cast<uint2 *>(mine + j0) = make_uint2(o0, o1);
                w = wn;
            }
            // the rest, the next piece in flight while one is worked on; NOT unrolled further: all 19 loads of a fully unrolled
            // loop were hoisted to its top (180 registers: two workgroups per CU instead of three)
#pragma unroll 2
            for (uint32_t j0 = 4u; j0 < NP; j0 += 2u) {
                const uint32_t jn = j0 + 2u < NP ? j0 + 2u : j0;
                const uint2 wn = *reinterpret_cast<const uint2 *>(mine + jn);
                const uint32_t o0 = pair(w.x, j0, std::false_type{}), o1 = pair(w.y, j0 + 1u, std::false_type{});
                if (EMIT) {
                    *reinterpret_cast<uint2 *>(mine + j0) = make_uint2(o0, o1);
                    last2 = o0; last1 = o1;
                }
                w = wn;
            }
            return V3{sx, sy, sz};
        };
        // pass 1, then the scan: inside the wavefront, then over the wavefronts
        V3 F = lo16(run(V3{tid == 0u ? xs.x : 0u, tid == 0u ? xs.y : 0u, tid == 0u ? xs.z : 0u}, std::false_type{}));
#pragma unroll
        for (int dd = 0; dd < 6; ++dd) {
            const M3 P = load_m3(itab + kRunTabPL + 9 * dd);  // A^(M 2^dd)
            const V3 up = shfl_up_v3(F, 1 << dd);
            if (lane >= (1 << dd)) F = lo16(add(F, mul(P, up)));
        }
        V3 E = shfl_up_v3(F, 1);  // the state in front of my run as far as my wavefront knows
        if (lane == 0) E = V3{0u, 0u, 0u};
        if (lane == 63) { s_F[wv][0] = F.x; s_F[wv][1] = F.y; s_F[wv][2] = F.z; }
        blk_barrier();
        V3 XW{0u, 0u, 0u};  // ... and in front of my wavefront
        if (NW > 1) {
            const M3 PW = load_m3(itab + kRunTabPL + 9 * 6);  // A^(64 M)
            for (int w = 0; w < wv; ++w) XW = lo16(add(mul(PW, XW), V3{s_F[w][0], s_F[w][1], s_F[w][2]}));
        }
        V3 S = E;  // (thread 0: the state in front of the block, as in pass 1)
        if (tid == 0u) { S.x = xs.x; S.y = xs.y; S.z = xs.z; }
        if (NW > 1 && wv > 0) S = lo16(add(mul(load_m3(itab + kRunTabPLANE + 9 * lane), XW), E));
        // pass 2: the samples
        (void)run(S, std::true_type{});
        // (up to seven samples lie behind the last lane's share when the block's first sample is not 16-byte aligned and the
        // buffer is full: the last lane goes on, one sample at a time)
        if (tid == NT - 1u && a0 + nsamp > NT * M) {
            V3 s{last1 >> 16, last1 & 0xffffu, last2 >> 16};
            for (uint32_t i = NT * M; i < a0 + nsamp; ++i) {
                const uint32_t v = (__umul24((uint32_t)obuf[i], sg) + __umul24(c1, s.x) + __umul24(c2, s.y) + __umul24(c3, s.z)) & 0xffffu;
                obuf[i] = (uint16_t)v;
                s = V3{v, s.x, s.y};
            }
        }
        blk_barrier();
        // the state behind these samples: the last three of them (fewer: what was in front moves down)
        if (tid == 0u) {
            const uint16_t *e = obuf + (a0 + nsamp);
            if (nsamp >= 3u) xs = V3{e[-1], e[-2], e[-3]};
            else if (nsamp == 2u) xs = V3{e[-1], e[-2], xs.x};
            else xs = V3{e[-1], xs.x, xs.y};
        }
    };
    if (unit >= total_units) return;
    RunRef cur = run_of(unit);
    uint4 img[NQ];
    if (cur.blk_lo < cur.blk_hi) fetch_image(cur, cur.blk_lo, img);
    for (;;) {
        // the next ticket, drawn a run ahead.  Inline asm: a returning atomic the compiler sees inside `if (tid == 0)` is
        // waited for at the end of that branch; this one is waited for where its value is used.  The address travels in a
        // VGPR pair (`off` form): the hardware interlocks VGPR operands, whereas an SGPR base restored from a spill by
        // v_readlane in the instruction in front needs five wait states that nothing pads inside an asm statement (round 2:
        // an experimental form of this kernel faulted on address 0 that way).  tools/check_asm_hazards.py, run by `make hip`
        // on the gfx950 disassembly, checks both that rule and that nothing touches next_ticket before the s_waitcnt below.
        uint32_t next_ticket = 0;
        if (tid == 0)
            asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(next_ticket) : "v"((uint64_t)(uintptr_t)ticket), "v"(1u) : "memory");
        BLK_STAMP(0);  // ticket
        const uint64_t g = cur.g, pay_lo = cur.pay_lo;
        const uint32_t n = cur.n, blk_lo = cur.blk_lo, blk_hi = cur.blk_hi;
        const WaveRef r = locate(G, g);
        const uint32_t len = r.len;
        int16_t *y = out + r.sample_off;
        const uint32_t n_blocks = (n + BG::kWords - 1u) / BG::kWords;
        RunRef nxt = cur;
        uint32_t next_unit = 0xffffffffu;
        if (blk_lo >= blk_hi) {  // an empty run (a waveform with fewer blocks than the longest): only the hand-over
            if (tid == 0) {
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(next_ticket)::"memory");
                s_next = next_ticket;
            }
            blk_barrier();
            next_unit = s_next;
            blk_barrier();
            if (next_unit < total_units) { nxt = run_of(next_unit); if (nxt.blk_lo < nxt.blk_hi) fetch_image(nxt, nxt.blk_lo, img); }
        }
        uint64_t run_base_c = 0;   // samples in front of the current block (known from the run's second block on)
        uint32_t run_acc = 0;      // the running sum there
        uint32_t carry_rel = 0;    // where the previous block's last code ended, in bits behind that block
        for (uint32_t blk = blk_lo; blk < blk_hi; ++blk) {
            const bool first_of_run = blk == blk_lo;
            const uint32_t sidx = (uint32_t)g * slots_per_wave + blk;  // this block's look-back entry and end word
            const uint32_t w0 = blk * BG::kWords;
            const uint32_t avail = (n - w0 < BG::kWords) ? n - w0 : BG::kWords;
            const uint32_t s_i0 = (uint32_t)((int64_t)(pay_lo + w0) - image_base(pay_lo, blk));  // image index of the block's first word
            store_image(cur, blk, img);
            if (tid == 0) s_defer = 0u;
            blk_barrier();
            BLK_STAMP(1);  // image

            // ---- phase 1: where the codes of my segment start, how many there are, what they sum to ----
            const uint32_t B0 = 32u * s_i0, bend = B0 + 32u * avail;
            const uint32_t bj = B0 + kSegBits * tid;
            const bool active = bj < bend;
            const uint32_t lim = (bj + kSegBits < bend) ? bj + kSegBits : bend;
            // lane 0 knows its start when the block is the waveform's first (bit 0) or follows one of this run
            const bool exact0 = tid == 0 && (blk == 0 || !first_of_run);
            const uint32_t start0 = B0 + (first_of_run ? 0u : carry_rel);
            uint32_t Qp = C - (exact0 ? start0 : (active ? bj - kBlkGuessBits : B0));
            uint32_t cnt = 0, sum = 0;
            blk_skip_pairs(W, k, active && !exact0, Qp, C - bj);
            if (!active) Qp = C - B0;
            uint32_t f = C - Qp;  // first code that starts in my segment
            uint32_t *const my_stage = stage + tid * kBlkLaneStride;
            // where the waveform's zero padding can be: its last payload word, if this block holds it
            const uint32_t qpad = (n - 1u >= w0 && n - 1u < w0 + BG::kWords) ? C - (B0 + 32u * (n - 1u - w0)) : 0u;
            if (qpad) blk_parse<kBlkCount, RESID, true>(W, k, active, Qp, C - lim, cnt, sum, 0u, nullptr, my_stage, qpad, kCap2);
            else blk_count_pairs<RESID>(W, k, active, Qp, C - lim, cnt, sum, my_stage, kCap2);
            if (!active) { cnt = 0; sum = 0; }
            uint32_t e = C - Qp;  // first code that starts behind it (or where the padding starts)
            BLK_STAMP(2);  // run-up + count (thread 0's wave)

            // Every lane must start where its predecessor ended; lanes that do not, start again from there.
            // CREEP: in a stream of equal-length codes whose pattern reads as the same codes from another phase (a slope-1 ramp
            // is "1010" per sample) a parse never falls into step: lanes that guessed the same wrong phase agree with one
            // another, and once lane 0 is put right the correction moves ONE lane per round (243 rounds per block measured,
            // NOPTREX-shaped ramps 81 ms).  Its signature -- the lowest lane that starts again advances by exactly one per
            // round and its end moves by the same amount each time -- is looked for, and after three such rounds every lane
            // behind the front is shifted by that amount at once.  Only where to start again is guessed (at most four times per
            // call); what is accepted is still the chain of equalities.  Noise and quiet data never show the signature.
            auto settle = [&]() __attribute__((always_inline)) {
                uint32_t prev_front = 0xffffffffu, creep = 0, jumps = 0;
                if (tid == 0) s_front = 0xffffffffu;
                for (uint32_t it = 0; it <= 2u * (uint32_t)NT + 8u; ++it) {
                    s_e[tid] = e;
                    blk_barrier();
                    const uint32_t want = tid ? s_e[tid - 1u] : f;
                    bool changed = active && want != f;
                    uint32_t from = want;
                    const uint32_t front = s_front;  // (lane << 8) | (how far its end moved + 128), 0xffffffff: nobody started again
                    if (it > 0u) {
                        const bool step = front != 0xffffffffu && prev_front != 0xffffffffu && (front >> 8) == (prev_front >> 8) + 1u &&
                                          (front & 0xffu) == (prev_front & 0xffu) && (front & 0xffu) != 128u && (front & 0xffu) != 0u;
                        creep = step ? creep + 1u : 0u;
                        prev_front = front;
                        if (creep >= 2u && jumps < 4u) {
                            creep = 0;
                            ++jumps;
                            prev_front = 0xffffffffu;
                            const uint32_t to = f + (front & 0xffu) - 128u;
                            if (active && tid > (front >> 8) && (int32_t)(to - bj) >= 0 && to < lim) { from = to; changed = from != f; }
                        }
                    }
                    if (!wg_any(changed)) break;  // (also: every read of s_e and s_front is done before the next write)
                    if (tid == 0) s_front = 0xffffffffu;
                    blk_barrier();
                    const uint32_t e_old = e;
                    if (changed) { f = from; Qp = C - f; cnt = 0; sum = 0; }
                    blk_parse<kBlkCount, RESID>(W, k, changed, Qp, C - lim, cnt, sum, 0u, nullptr, my_stage, qpad, kCap2);
                    if (changed) {
                        e = C - Qp;
                        const int32_t d = (int32_t)(e - e_old);
                        const uint32_t dd = (d > -128 && d < 128) ? (uint32_t)(d + 128) : 0u;
                        __hip_atomic_fetch_min(&s_front, (tid << 8) | dd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            };
            // A block whose lane 0 only guessed its start and reads a pattern that another phase reads as the same codes does
            // not publish its end before its predecessor's has confirmed the guess: the end would move, and a successor that
            // started from it sends the whole waveform to the one-workgroup fallback (ramps: 25 x 14 M samples 824 ms).
            // (asked of lane 0, whose guess decides whether the block has to start again, and of the last lane, whose end is the one published)
            if (first_of_run && blk > 0u && cnt >= 8u && (tid == 0u || tid == (avail + (uint32_t)kBlkSegW - 1u) / (uint32_t)kBlkSegW - 1u)) {
                bool amb = false;
                const BlkPair p0 = blk_pair<false>(W, k, C - f);
                const uint32_t f2 = f + (0u - p0.nu1), P = 0u - p0.nu2;  // (nu = minus the code length; from the second code on)
                if (P != 0u && P < 26u && P * (cnt - 1u) == e - f2) {  // codes of one length ...
                    const uint32_t w0 = blk_window(W, C - f2);
                    if (w0 == blk_window(W, C - (f2 + P)) && w0 == blk_window(W, C - (f2 + 2u * P))) {  // ... of one pattern ...
                        for (uint32_t r = 1; r < P && !amb; ++r)  // ... that reads as a code of that length from another phase
                            amb = (0u - blk_pair<false>(W, k, C - (f2 + r)).nu1) == P;
                    }
                }
                if (amb) s_defer = 1u;
            }
            settle();
            const bool defer = s_defer != 0u;  // (read behind settle()'s barriers)
            BLK_STAMP(3);  // settle (includes waiting for the slowest wave)
            const uint32_t last_active = (avail + (uint32_t)kBlkSegW - 1u) / (uint32_t)kBlkSegW - 1u;
            const uint32_t e_last0 = s_e[last_active];
            const bool last_of_run = blk + 1u == blk_hi;
            // the end of this run = the start of the next one, published as soon as it is known
            if (tid == 0 && last_of_run && !defer)
                __hip_atomic_store(ends + sidx, 0x80000000u | (e_last0 - bend), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (first_of_run && blk > 0) {
                if (tid == 0) {
                    uint32_t v = 0, spins = 0;
                    for (;;) {
                        v = __hip_atomic_load(ends + sidx - 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (v & 0x80000000u) break;
                        __builtin_amdgcn_s_sleep(2);
                        if (++spins > (1u << 24)) { atomicOr(&st->err, kErrInternal); break; }  // cannot happen; never hang
                    }
                    s_pred = v & 0xffffu;
                }
                blk_barrier();
                const uint32_t true_f0 = B0 + s_pred;
                const bool fix0 = tid == 0 && true_f0 != f;
                if (fix0) s_next = true_f0 - f;  // (how far lane 0 moves; s_next is not in use here)
                if (wg_any(fix0)) {
                    // a block that held its end back reads one pattern throughout: its lanes all guessed the phase lane 0 guessed,
                    // and move with it (a guess again: settle() below accepts nothing but the chain of equalities)
                    bool mv = fix0;
                    uint32_t to = true_f0;
                    if (defer && tid != 0u && active) {
                        const uint32_t t = f + s_next;
                        if ((int32_t)(t - bj) >= 0 && t < lim) { mv = true; to = t; }
                    }
                    if (mv) { f = to; Qp = C - f; cnt = 0; sum = 0; }
                    blk_parse<kBlkCount, RESID>(W, k, mv, Qp, C - lim, cnt, sum, 0u, nullptr, my_stage, qpad, kCap2);
                    if (mv) e = C - Qp;
                    settle();
                    // a one-block run has published its end already, and its successor has started from it: if that end
                    // moved, the successor's run is wrong
                    if (tid == 0 && last_of_run && !defer && s_e[last_active] != e_last0 && blk + 1u < n_blocks) atomicExch(fail + g, 1u);
                }
                if (tid == 0 && last_of_run && defer)
                    __hip_atomic_store(ends + sidx, 0x80000000u | (s_e[last_active] - bend), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            const uint32_t e_end = s_e[last_active];  // (after a correction: the corrected end)
            BLK_STAMP(4);  // predecessor's end

            // ---- samples and residual sum in front of my segment (workgroup scan) and in front of the block ----
            const uint32_t incl_c = wave_incl_scan_dpp(cnt), incl_s = wave_incl_scan_dpp(sum);
            if (lane == 63) { s_tot[0][wv] = incl_c; s_tot[1][wv] = incl_s; }
            if (tid == 0 && last_of_run) {  // (drawn at the start of the run: it has arrived)
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(next_ticket)::"memory");
                s_next = next_ticket;
            }
            blk_barrier();
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
                uint64_t ex_c = run_base_c, ex_s = run_acc;
                if (blk == 0) {
                    ex_c = 0;
                    ex_s = 0;
                } else if (first_of_run) {
                    // decoupled look-back over the blocks of the waveform in front of this one
                    ex_c = 0;
                    ex_s = 0;
                    if (lane == 0) __hip_atomic_store(state + sidx, kScanAgg | mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const int64_t first = (int64_t)sidx - (int64_t)blk;  // block 0 of this waveform
                    int64_t base = (int64_t)sidx - 1;
                    uint32_t spins = 0;
#ifndef DRX_BLK_NO_GATE
                    // the nearest predecessor alone first (one 8-byte load per poll, not a window from every waiting workgroup)
                    for (;;) {
                        uint64_t v = 0;
                        if (lane == 0) v = __hip_atomic_load(state + base, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 62)) != 0) break;
                        __builtin_amdgcn_s_sleep(4);
                        if (++spins > (1u << 22)) break;  // (the window loop below reports it)
                    }
#endif
                    for (;;) {
                        // lane l looks at predecessors base - l (nearer) and base - 64 - l (farther)
                        const int64_t i0 = base - lane, i1 = base - 64 - lane;
                        uint64_t s0v = kScanPrefix, s1v = kScanPrefix;  // in front of block 0: an empty prefix
                        if (i0 >= first) s0v = __hip_atomic_load(state + i0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (i1 >= first) s1v = __hip_atomic_load(state + i1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const uint32_t st0 = (uint32_t)(s0v >> 62), st1 = (uint32_t)(s1v >> 62);
                        const uint64_t p0 = __ballot(st0 == 2u), z0 = __ballot(st0 == 0u);
                        const uint64_t p1 = __ballot(st1 == 2u), z1 = __ballot(st1 == 0u);
                        const int fp = p0 ? __builtin_ctzll(p0) : (p1 ? 64 + __builtin_ctzll(p1) : 128);  // nearest prefix
                        const uint64_t near0 = fp >= 64 ? ~0ull : ((1ull << fp) - 1ull);
                        const uint64_t near1 = fp >= 128 ? ~0ull : (fp > 64 ? ((1ull << (fp - 64)) - 1ull) : 0ull);
                        if ((z0 & near0) | (z1 & near1)) {  // a nearer predecessor has not published yet
                            __builtin_amdgcn_s_sleep(1);
                            if (++spins > (1u << 22)) { if (lane == 0) atomicOr(&st->err, kErrInternal); break; }
                            continue;
                        }
                        const uint64_t v0 = (lane <= fp) ? (s0v & kScanValMask) : 0ull;
                        const uint64_t v1 = (64 + lane <= fp) ? (s1v & kScanValMask) : 0ull;
                        ex_c += wave_sum_u64((v0 >> 16) + (v1 >> 16));
                        ex_s += wave_sum_u64((v0 & 0xffffull) + (v1 & 0xffffull));
                        if (fp < 128) break;
                        base -= 128;
                    }
                }
                // everything in front of this block is known now: its successors find a prefix here
                if (lane == 0)
                    __hip_atomic_store(state + sidx, kScanPrefix | ((((ex_c + tot_c) << 16) | ((ex_s + tot_s) & 0xffffull)) & kScanValMask),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (lane == 0) { s_b[0] = ex_c; s_b[1] = ex_s; }
            }
            blk_barrier();
            BLK_STAMP(5);  // scan + look-back
            const uint64_t base_c = s_b[0];
            const uint32_t acc_base = (uint32_t)s_b[1];
            run_base_c = base_c + tot_c;
            run_acc = acc_base + tot_s;
            carry_rel = e_end - bend;
            // The next block's image -- of this run or of the next ticket's -- travels while this one's samples are put in
            // order and written out.  (Issued earlier -- at the top of the block, or behind the count -- the loads were
            // measured no faster to arrive: the wait in front of the next block is for this block's output lines, whose
            // store loop the compiler cannot count, and retiring the loads by hand in front of those stores only moved the
            // wait there: profiles/r02_notes.md.)
            if (!last_of_run) {
                fetch_image(cur, blk + 1u, img);
            } else {
                next_unit = s_next;
                if (next_unit < total_units) { nxt = run_of(next_unit); if (nxt.blk_lo < nxt.blk_hi) fetch_image(nxt, nxt.blk_lo, img); }
            }
            // the waveform has `len` samples; a code decoded out of the zero padding behind the last one does not count
            const uint32_t blk_first = base_c < (uint64_t)len ? (uint32_t)base_c : len;
            const uint32_t blk_count = (tot_c < len - blk_first) ? tot_c : len - blk_first;
            const uint32_t rel0 = pre_c + incl_c - cnt;  // my first sample, relative to the block's first
            const uint32_t todo = rel0 >= blk_count ? 0u : ((cnt < blk_count - rel0) ? cnt : blk_count - rel0);
            // Verdicts, left to the kernel that runs after this one (a block behind a mis-started one counts garbage, and its
            // waveform is flagged for the fallback anyway): the waveform's last block must bring the count to exactly `len`
            // samples, and its last code must end in the last payload word: n_i = ceil(bits / 32) (src/deltaRice.c:237-241).
            if (tid == 0 && blk + 1u == n_blocks) {
                if (base_c + tot_c != (uint64_t)len || w0 + ((e_end - B0 + 31u) >> 5) != n) atomicExch(suspect + g, 1u);
            }

            // FUSE: the filter's state in front of this block -- zero at the waveform's start, thread 0's own inside a run, the
            // previous run's last block's across runs (long published where runs of one waveform are not in flight together)
            if (FUSE && first_of_run && tid == 0u) {
                xs = V3{0u, 0u, 0u};
                if (blk > 0u) {
                    uint64_t v = 0;
                    uint32_t spins = 0;
                    for (;;) {
                        v = __hip_atomic_load(xstate + sidx - 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (v >> 63) break;
                        __builtin_amdgcn_s_sleep(2);
                        if (++spins > (1u << 24)) { atomicOr(&st->err, kErrInternal); break; }  // cannot happen; never hang
                    }
                    xs = V3{(uint32_t)v & 0xffffu, (uint32_t)(v >> 16) & 0xffffu, (uint32_t)(v >> 32) & 0xffffu};
                }
            }

            // ---- phase 2: the samples in output order, whole lines to HBM ----
            const uint32_t a0 = (uint32_t)((((uintptr_t)(y + blk_first)) >> 1) & 7u);  // kOutCap is a multiple of 8: the same every pass
            auto copy_out = [&](uint32_t R0) __attribute__((always_inline)) {  // staged samples [R0, R0 + kOutCap) of the block -> HBM
                const uint32_t nsamp = (blk_count - R0 < BG::kOutCap) ? blk_count - R0 : BG::kOutCap;
                g_u16 *gbase = (g_u16 *)(y + blk_first + R0) - a0;  // 16-byte aligned
                // whole 16-byte pieces [p_lo, p_hi) without a test per piece; the up to seven samples in front of the first and
                // behind the last one by sixteen lanes
                const uint32_t end = a0 + nsamp, p_lo = (a0 + 7u) >> 3, p_hi = end >> 3;
                auto piece = [&](uint32_t p) __attribute__((always_inline)) {
                    const uint4 v = *reinterpret_cast<const uint4 *>(obuf + 8u * p);
                    *(g_uint4 *)(gbase + 8u * p) = (u32x4v){v.x, v.y, v.z, v.w};
                };
                uint32_t p = tid;  // (piece p by thread p mod NT: a wavefront's store instruction covers whole aligned lines)
                if (p >= p_lo && p < p_hi) piece(p);
                for (p += NT; p < p_hi; p += NT) piece(p);
                if (tid < 16u) {
                    const uint32_t s = tid < 8u ? tid : 8u * p_hi + (tid - 8u);
                    const bool ok = tid < 8u ? (s >= a0 && s < 8u * p_lo && s < end) : (p_hi >= p_lo && s >= a0 && s < end);
                    if (ok) gbase[s] = obuf[s];
                }
            };
            // the common case: every lane's codes are all samples of the waveform and fit its share of the staging buffer
            const bool lane_ok = cnt <= kBlkLaneCap && todo == cnt;
            if (!wg_any(!lane_ok)) {
                constexpr int NR = (int)(kBlkLaneCap / 2u);
                uint32_t rr[NR];
                // (pairs beyond the wavefront's largest count are skipped by a scalar branch: 54 of the 76 slots are used on average)
                const uint32_t wmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_max_u32(cnt));
#pragma unroll
                for (int i = 0; i < NR; ++i) {
                    rr[i] = 0;
                    if (2u * (uint32_t)i < wmax) rr[i] = my_stage[i];
                }
#ifndef DRX_BLK_REORDER16
                if (RESID) s_e[tid] = sum;  // (a lane's last residual: what the next lane puts in front of its first one, below)
#endif
                blk_barrier();  // every lane holds its samples: the buffer may now be rewritten in output order
                const uint32_t base16 = RESID ? 0u : (acc_base + pre_s + incl_s - sum) & 0xffffu;  // the running sum in front of my first sample
                // both halves of a dword take the base in one packed add; the two 16-bit stores have immediate offsets from one
                // address and are masked by the lane's count (an exec mask costs a compare, a dump slot a compare and a select)
                typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
                const u16x2 b2 = {(uint16_t)base16, (uint16_t)base16};
                uint16_t *const op = obuf + (a0 + rel0);
#ifndef DRX_BLK_REORDER16
                {
                    // WHOLE dwords: a lane whose first sample sits in the high half of a dword (odd position) writes that dword
                    // with the sample in front of its first one in the low half -- which is the running sum in front of it, base16,
                    // whoever decoded it (residual mode: the lane in front's last residual, through s_e) -- and a lane whose last
                    // sample sits in a low half leaves the high half to its successor and stores that sample alone, once (both
                    // stores carry the same value).  38 four-byte stores per lane instead of 76 two-byte ones, at the same three
                    // VALU instructions per pair (add, byte permute, compare).
                    const uint32_t o = a0 + rel0;
                    const bool odd = (o & 1u) != 0u;
                    uint32_t *const dp = reinterpret_cast<uint32_t *>(obuf) + (o >> 1);
                    const uint32_t sel = odd ? 0x05040302u : 0x07060504u;  // odd: (prev.hi, cur.lo); even: cur
                    const uint32_t lim = cnt == 0u ? 0u : (odd ? cnt : cnt - 1u);  // dword j is written if 2 j < lim
                    uint32_t prevp = RESID ? (tid ? s_e[tid - 1u] << 16 : 0u) : __builtin_bit_cast(uint32_t, b2);
#pragma unroll
                    for (int i = 0; i < NR; ++i) {
                        if (2u * (uint32_t)i < wmax) {
                            const uint32_t cur = __builtin_bit_cast(uint32_t, (u16x2)(__builtin_bit_cast(u16x2, rr[i]) + b2));
                            const uint32_t d = __builtin_amdgcn_perm(cur, prevp, sel);
                            if (2u * (uint32_t)i < lim) dp[i] = d;
                            prevp = cur;
                        }
                    }
                    if (cnt != 0u) op[cnt - 1u] = (uint16_t)(base16 + sum);  // (my last sample = the running sum behind my codes; residual mode: my last residual)
                }
#else
#pragma unroll
                for (int i = 0; i < NR; ++i) {
                    if (2u * (uint32_t)i < wmax) {
                        const u16x2 v = __builtin_bit_cast(u16x2, rr[i]) + b2;
                        if (2u * (uint32_t)i < cnt) op[2 * i] = v.x;
                        if (2u * (uint32_t)i + 1u < cnt) op[2 * i + 1] = v.y;
                    }
                }
#endif
                blk_barrier();
                BLK_STAMP(6);  // reorder
                if (FUSE) iir_lds(a0, blk_count);
                copy_out(0u);
                BLK_STAMP(7);  // copy-out
            } else {
                // some lane holds more codes than its share (long runs of tiny residuals), or codes past the waveform's
                // last sample (a corrupt stream): decode again from f, in as many staging passes as the block needs
                uint32_t c = 0, acc = acc_base + pre_s + incl_s - sum;
                Qp = C - (todo ? f : B0);
                for (uint32_t R0 = 0; R0 < blk_count; R0 += BG::kOutCap) {
                    // my samples with block-relative index below R0 + kOutCap
                    const uint32_t cmax = (rel0 >= R0 + BG::kOutCap) ? 0u : ((todo < R0 + BG::kOutCap - rel0) ? todo : R0 + BG::kOutCap - rel0);
                    // slot of sample c: a0 + rel0 + c - R0 (>= a0 for every c this pass decodes)
                    uint16_t *outp = obuf + (int32_t)(a0 + rel0 - R0);
                    blk_parse<kBlkValue, RESID>(W, k, c < cmax, Qp, 0u, c, acc, cmax, outp);
                    blk_barrier();
                    if (FUSE) iir_lds(a0, (blk_count - R0 < BG::kOutCap) ? blk_count - R0 : BG::kOutCap);
                    copy_out(R0);
                    blk_barrier();
                }
            }
            if (FUSE && last_of_run && tid == 0u)
                __hip_atomic_store(xstate + sidx, (1ull << 63) | (uint64_t)(xs.x & 0xffffu) | ((uint64_t)(xs.y & 0xffffu) << 16) | ((uint64_t)(xs.z & 0xffffu) << 32),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            blk_barrier();  // W, the staging buffer and the s_* words are rewritten by the next block
            BLK_STAMP(8);
#ifdef DRX_BLK_STAMPS
            if (tid == 0) s_prof[9] += 1ull;
#endif
        }
        if (next_unit >= total_units) { flush_prof(); return; }
        unit = next_unit;
        cur = nxt;
    }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
static int nt_for_len(uint32_t wave_len, uint32_t k) {
    // lanes per block: a waveform of about (k + 3.5) bits per sample should fill most of its last block
#ifdef DRX_BLK_FORCE_NT
    return DRX_BLK_FORCE_NT;
#endif
    const uint64_t typ_words = ((uint64_t)wave_len * (2u * k + 7u)) >> 6;
    return typ_words <= blk_words(64, k) ? 64 : (typ_words <= blk_words(128, k) ? 128 : 256);
}
static int blocks_nt(const Geom &G) { return G.uniform ? nt_for_len(G.u_wave_len, G.k) : (int)G.rag_blk_nt; }

// blocks of `waves` waveforms of `wave_len` samples, weighted by what a block of theirs costs (1.7 for waveforms of one or two
// blocks: they pay the ticket's dependent loads -- ticket, table entry, image -- per block: 36 us measured at 256 lanes, 10
// chunks of 1166 x 12 000 and of 854 x 16 384)
static double blocks_weighted(uint64_t waves, uint32_t wave_len, uint32_t k, int nt) {
    const uint64_t typ_words = ((uint64_t)wave_len * (2u * k + 7u)) >> 6;
    const uint64_t bpw = (typ_words + blk_words((uint32_t)nt, k) - 1u) / blk_words((uint32_t)nt, k);
    return (double)(waves * bpw) * (bpw <= 2u ? 1.7 : 1.0);
}
static double blocks_us_of(double weighted_blocks, int nt, uint32_t k) {
    const double resident = nt == 256 ? 768.0 : (nt == 128 ? 1536.0 : 3072.0);
    // (measured with 11 words per lane; a block's time goes with its bits)
    const double t_blk = (nt == 256 ? 21.0 : (nt == 128 ? 28.0 : 41.0)) * (double)blk_segw_plan(k) / 11.0;
    return (double)(uint64_t)((weighted_blocks + resident - 1.0) / resident) * t_blk;  // whole rounds of the resident grid
}

// Which batches take this decoder: the delta filter or a general filter the fast kernels take (its inverse then runs in place
// behind this decoder, drx_iir.hip), waveforms of at least 2048 samples, and cheaper here than a lane per waveform.  The two
// costs (tools/len_sweep.py, profiles/r02_blocks_vs_lanes.txt, r02_len_sweep_*): a lane decodes ~17 samples per microsecond
// whatever else runs, so k_decode_lanes takes ~60 ns x WaveformLength per 98 304 waveforms; a block costs a workgroup 21 /
// 28 / 41 us at 256 / 128 / 64 lanes, full or not, with 768 / 1536 / 3072 workgroups resident.  (Round 2's first rule, "at
// most 24 576 waveforms", sent 25 chunks of 854 x 16 384 here: two blocks per waveform, the second a fifth full, 1.46 ms
// where the lane kernel takes 0.98.)  Ragged batches: decided once from the host's chunk table, blocks_plan_ragged().
bool blocks_batch(const Geom &G) {
    if (G.n_taps != 0 && !G.fast_taps) return false;
    if (!G.uniform) return G.rag_blocks != 0;
    if (!(G.total_waves <= 98304u && G.u_wave_len >= 2048u)) return false;
    const int nt = blocks_nt(G);
    const uint64_t typ_words = ((uint64_t)G.u_wave_len * (2u * G.k + 7u)) >> 6;
    const uint64_t bpw = (typ_words + blk_words((uint32_t)nt, G.k) - 1u) / blk_words((uint32_t)nt, G.k);
    const double blocks_us = blocks_us_of((double)(G.total_waves * bpw), nt, G.k) * (bpw <= 2u ? 1.7 : 1.0);
    const double lanes_us = 0.06 * (double)G.u_wave_len * (double)((G.total_waves + 98303u) / 98304u);
    return blocks_us < lanes_us;
}

void blocks_plan_ragged(Geom &G, const ChunkDesc *d, uint32_t *list_out) {
    G.rag_blocks = 0;
    G.rag_blk_classes = 0;
    uint32_t max_len = 0, min_len = 0xffffffffu;
    for (uint64_t c = 0; c < G.n_chunks; ++c) {
        max_len = d[c].wave_len > max_len ? d[c].wave_len : max_len;
        min_len = d[c].wave_len < min_len ? d[c].wave_len : min_len;
    }
    if (G.total_waves > 98304u || min_len < 2048u) return;
    const int nt = nt_for_len(max_len, G.k);
    double wb = 0;
    for (uint64_t c = 0; c < G.n_chunks; ++c) wb += blocks_weighted(d[c].n_waves, d[c].wave_len, G.k, nt);
    // a lane takes 60 ns per sample: a lane-per-waveform launch lasts as long as its longest waveform, per 98 304 of them
    const double lanes_us = 0.06 * (double)max_len * (double)((G.total_waves + 98303u) / 98304u);
    if (!(blocks_us_of(wb, nt, G.k) < lanes_us)) return;
    // look-back slots per waveform: enough for the longest one at 25 bits per sample, in blocks of the SMALLEST class's size
    const uint64_t per = (max_payload_words(max_len) + blk_words_min(64u) - 1u) / blk_words_min(64u);
    if (G.total_waves * per * 12u > (1ull << 30)) return;  // (one very long waveform among very many: the table would not pay)
    G.rag_blocks = 1u;
    G.rag_blk_nt = (uint32_t)nt;
    G.rag_blk_slots = (uint32_t)(per ? per : 1u);
    // classes by floor(log2 WaveformLength), longest first (the long classes start while the grid is empty)
    uint32_t n_cls = 0, at = 0;
    for (int b = 31; b >= 11; --b) {
        uint32_t cnt = 0, cls_max = 0;
        for (uint64_t c = 0; c < G.n_chunks; ++c) {
            if ((31 - __builtin_clz(d[c].wave_len)) != b) continue;
            for (uint32_t i = 0; i < d[c].n_waves; ++i) list_out[at + cnt++] = (uint32_t)(d[c].wave_base + i);
            cls_max = d[c].wave_len > cls_max ? d[c].wave_len : cls_max;
        }
        if (!cnt) continue;
        G.rag_blk_class_off[n_cls] = at;
        G.rag_blk_class_len[n_cls] = cls_max;
        at += cnt;
        ++n_cls;
    }
    G.rag_blk_class_off[n_cls] = at;
    G.rag_blk_classes = n_cls;
}

// the fused inverse filter's tables: one set per lane share (76, 68, 60 samples), kRunTabWords each
uint32_t blocks_iir_tab_words() { return 3u * kRunTabWords; }
void blocks_iir_tables(const uint32_t fast_nt[3], uint32_t t0neg, uint32_t *tab) {
    iir_run_tables(fast_nt, t0neg, 76u, tab);
    iir_run_tables(fast_nt, t0neg, 68u, tab + kRunTabWords);
    iir_run_tables(fast_nt, t0neg, 60u, tab + 2u * kRunTabWords);
}

static uint32_t blocks_slots_per_wave(const Geom &G) {  // blocks of a waveform at 25 bits per sample
    if (!G.uniform) return G.rag_blk_slots;
    const uint64_t per = (max_payload_words(G.u_wave_len) + blk_words_min(blocks_nt(G)) - 1u) / blk_words_min(blocks_nt(G));
    return (uint32_t)(per ? per : 1u);
}

constexpr uint32_t kBlkMaxClasses = 32;  // ragged batches: one launch per class of WaveformLengths floor(log2 L)
// scratch: u32 info[32][4] | u32 fail[W] | u32 suspect[W] | u32 ticket[1] (+ pad to 16 bytes) | u32 ends[slots] | u64 state[slots] | u64 xstate[slots] | prof
struct BlkScratch {
    uint32_t *info, *fail, *suspect, *ticket, *ends;
    uint64_t *state, *xstate;  // xstate: the inverse filter's state behind every block slot (FUSE)
    unsigned long long *prof;  // 16 counters of the diagnostic build
    uint64_t bytes;
};
static BlkScratch blocks_layout(const Geom &G, void *base) {
    const uint64_t W = G.total_waves, U = W * blocks_slots_per_wave(G);
    uint64_t n32 = 4u * kBlkMaxClasses + W + W + 1u;  // info[class][4]: {most blocks of a waveform, tickets, the class's ticket word, -}
    n32 = (n32 + 3u) & ~3ull;
    BlkScratch L;
    uint32_t *p = reinterpret_cast<uint32_t *>(base);
    L.info = p;
    L.fail = p + 4u * kBlkMaxClasses;
    L.suspect = L.fail + W;
    L.ticket = L.suspect + W;
    L.ends = p + n32;
    const uint64_t ends32 = (U + 1u) & ~1ull;
    L.state = reinterpret_cast<uint64_t *>(L.ends + ends32);
    L.xstate = L.state + U;
    L.prof = reinterpret_cast<unsigned long long *>(L.xstate + U);
    L.bytes = (n32 + ends32) * 4u + 2u * U * 8u + 16u * 8u;
    return L;
}

uint64_t blocks_scratch_bytes(const Geom &G) { return blocks_batch(G) ? blocks_layout(G, nullptr).bytes : 0; }

hipError_t launch_decode_blocks(const Geom &G, const uint32_t *d_in, uint64_t in_words, const uint64_t *d_wave_off,
                                const uint32_t *d_wave_words, void *d_blk, DevStatus *d_status, int16_t *d_out,
                                const uint32_t **fail_out, const uint32_t **suspect_out, bool resid, hipStream_t s, bool *fused_out) {
    const BlkScratch L = blocks_layout(G, d_blk);
    hipError_t e = hipMemsetAsync(d_blk, 0, L.bytes, s);
    if (e != hipSuccess) return e;
    // resident grid: 256 CUs x workgroups per CU (LDS: 51 KB at NT = 256, 26 KB at 128, 13 KB at 64), never more than there are units
    const uint32_t spw = blocks_slots_per_wave(G);
    // the block geometry of this decode: by the stream's bits per sample as the caller states them (a wrong in_words costs speed, nothing else)
    const uint64_t b10 = G.total_samples ? 320ull * in_words / G.total_samples : 65ull;
#ifdef DRX_BLK_FORCE_SW
    const int sw = DRX_BLK_FORCE_SW;
#else
    const int sw = blk_segw_bits10(b10);
#endif
    // General filters: the inverse filter inside this kernel (FUSE) where its state can pass from a run's last block to the next
    // run's first without a chain of waits, i.e. where every launch has at least as many waveforms as resident workgroups (the
    // condition under which runs are longer than one block); else residuals now and k_iir_tiles behind (debug flag 2097152: always).
    auto resident_of = [](int nt) { return 256u * (nt == 64 ? 12u : (nt == 128 ? 6u : 3u)); };
    // (General filters run at 128 or 256 lanes per block when fused and at 256 when not: 12 of the 36 instantiations of the
    // kernel are not built; a block larger than the class's choice costs short waveforms some empty lanes, nothing else.)
    auto nt_gen = [](int nt) { return nt < 128 ? 128 : nt; };
    bool fuse = resid && G.blk_iir_tab != nullptr && !(G.dbg & 2097152u);
    if (fuse) {
        if (G.uniform) fuse = G.total_waves >= resident_of(nt_gen(blocks_nt(G)));
        else
            for (uint32_t c = 0; c < G.rag_blk_classes; ++c)
                fuse = fuse && (G.rag_blk_class_off[c + 1] - G.rag_blk_class_off[c]) >= resident_of(nt_gen(nt_for_len(G.rag_blk_class_len[c], G.k)));
    }
    if (fused_out) *fused_out = fuse;
    auto launch_class = [&](uint32_t cls, const uint32_t *list, uint32_t n_waves, int nt, uint32_t wave_len) {
        if (resid) nt = fuse ? nt_gen(nt) : 256;
        uint32_t *info = L.info + 4u * cls;
        const uint32_t words_per_block = (uint32_t)nt * (uint32_t)sw;
        k_blk_max<<<1, 1024, 0, s>>>(n_waves, d_wave_words, words_per_block, info, list);
        const uint64_t units = (uint64_t)n_waves * spw;
        // Runs of several blocks only when two runs of one waveform are RARELY in flight together (see the kernel): at least
        // as many waveforms as resident workgroups.  (Ticket order does not exclude it -- a slow workgroup holding ticket t may
        // still run when ticket t + n_waves is drawn; the later run's look-back then simply waits on the lower ticket.)  Then as long as the launch keeps kBlkRounds tickets per workgroup (the
        // tail of the last round), up to the whole waveform: only a run's first block waits for other workgroups (nEDM, 6
        // blocks per waveform: one run; NOPTREX, 36: three runs of 12: 1.40 / 0.98 ms against 1.43 / 1.00 with round 2's fixed 4).
        const uint32_t resident = resident_of(nt);
        const uint64_t typ_words = ((uint64_t)wave_len * b10) / 320u;
        const uint64_t bpw = (typ_words + words_per_block - 1u) / words_per_block;
        uint64_t rl = ((uint64_t)n_waves * bpw) / ((uint64_t)kBlkRounds * resident);
        rl = rl > bpw ? bpw : rl;
        const uint32_t run_len = n_waves >= resident ? (uint32_t)(rl < 1u ? 1u : rl) : 1u;
        auto go = [&](auto nt_tag, auto resid_tag, auto sw_tag, unsigned per_cu) {
            constexpr int NT = decltype(nt_tag)::value, SW = decltype(sw_tag)::value;
            constexpr int MODE = decltype(resid_tag)::value;  // 0 delta filter, 1 residuals (k_iir_tiles follows), 2 inverse filter fused
            const unsigned grid = (unsigned)(units < 256u * per_cu ? units : 256u * per_cu);
            // the filter's tables for runs of this geometry's lane share (76 / 68 / 60 samples)
            const uint32_t *itab = G.blk_iir_tab ? G.blk_iir_tab + kRunTabWords * (blk_lane_cap(SW) == 76u ? 0u : (blk_lane_cap(SW) == 68u ? 1u : 2u)) : nullptr;
            k_decode_blocks<NT, MODE != 0, SW, MODE == 2><<<grid, NT, 0, s>>>(G, d_in, in_words, d_wave_off, d_wave_words, info, spw, run_len, L.state, L.ends,
                                                                             info + 2, L.fail, L.suspect, d_status, d_out, L.prof, list, n_waves, itab, L.xstate);
        };
        auto by_sw = [&](auto nt_tag, auto resid_tag, unsigned per_cu) {
            switch (sw) {
                case 9: go(nt_tag, resid_tag, std::integral_constant<int, 9>{}, per_cu); break;
                case 11: go(nt_tag, resid_tag, std::integral_constant<int, 11>{}, per_cu); break;
                case 15: go(nt_tag, resid_tag, std::integral_constant<int, 15>{}, per_cu); break;
                default: go(nt_tag, resid_tag, std::integral_constant<int, 19>{}, per_cu); break;
            }
        };
        auto by_resid = [&](auto nt_tag, unsigned per_cu) {
            if (fuse) by_sw(nt_tag, std::integral_constant<int, 2>{}, per_cu);
            else if (resid) by_sw(nt_tag, std::integral_constant<int, 1>{}, per_cu);
            else by_sw(nt_tag, std::integral_constant<int, 0>{}, per_cu);
        };
        // resident workgroups per CU by LDS: 50 dwords per lane in every class
        auto delta_only = [&](auto nt_tag, unsigned per_cu) { by_sw(nt_tag, std::integral_constant<int, 0>{}, per_cu); };
        if (nt == 64) delta_only(std::integral_constant<int, 64>{}, 12u);
        else if (nt == 128) { if (fuse) by_sw(std::integral_constant<int, 128>{}, std::integral_constant<int, 2>{}, 6u); else delta_only(std::integral_constant<int, 128>{}, 6u); }
        else by_resid(std::integral_constant<int, 256>{}, 3u);
    };
    if (G.uniform) {
        launch_class(0u, nullptr, (uint32_t)G.total_waves, blocks_nt(G), G.u_wave_len);
    } else {
        for (uint32_t c = 0; c < G.rag_blk_classes; ++c)
            launch_class(c, G.rag_blk_list + G.rag_blk_class_off[c], G.rag_blk_class_off[c + 1] - G.rag_blk_class_off[c],
                         nt_for_len(G.rag_blk_class_len[c], G.k), G.rag_blk_class_len[c]);
    }
#ifdef DRX_BLK_STAMPS
    {
        unsigned long long h[16];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(h, L.prof, sizeof h, hipMemcpyDeviceToHost);
        static const char *names[9] = {"ticket", "image load", "run-up + count", "settle", "pred end", "scan + look-back", "reorder", "copy-out", "tail"};
        unsigned long long tot = 0;
        for (int i = 0; i < 9; ++i) tot += h[i];
        fprintf(stderr, "[blk stamps]");
        for (int i = 0; i < 9; ++i) fprintf(stderr, " %s %.1f%%", names[i], 100.0 * (double)h[i] / (double)(tot ? tot : 1));
        fprintf(stderr, "  | %llu blocks, %.0f ticks of s_memtime each\n", h[9], (double)tot / (double)(h[9] ? h[9] : 1));
    }
#endif
    *fail_out = L.fail;
    *suspect_out = L.suspect;
    return hipGetLastError();
}

}  // namespace drx
