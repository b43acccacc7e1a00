/*
 * deltarice_oracle.c -- CPU restatement of the Delta-Rice chunk codec.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity checker for the HIP path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * call it.  Nothing under deltarice_amd/ links, loads or imports it: the
 * product path is the HIP library and fails loudly when that is missing.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function
 * here against (a) the known-answer vector of the reference's docs
 * (docs/Algorithm.md:9) and (b) tests/golden/*.npz, whose expected bytes were
 * produced by the reference's own src/deltaRice.c compiled unmodified into
 * oracle/_ref/ (recipe: oracle/Makefile, generator: tests/golden/make_golden.py).
 *
 * The algorithm is restated from the on-disk format, not transliterated:
 *   chunk   := u32 N | { u32 n_i | u32 payload_i[n_i] } for each waveform i
 *              (reference: src/deltaRice.c:415,379,427-433)
 *   payload := MSB-first concatenation of one code per sample, last word
 *              left-aligned and zero padded            (src/deltaRice.c:229-241)
 *   code(z) := z>>k zeros, '1', k low bits of z         if (z>>k) < 8
 *              8 zeros, '1', 16 bits of z               otherwise  (:212-228)
 *   z       := zig-zag of the prediction residual d     (:207-211)
 *   d       := causal FIR of the waveform, mod 2^16     (:49-76), default taps
 *              [1,-1] = delta encoding; decode inverts it as an IIR (:78-103)
 *
 * Behaviour follows the reference's OpenMP build (the canonical one: it is the
 * build that handles a trailing partial waveform, src/deltaRice.c:420-425,
 * 329-334).  Deliberate differences, all on inputs where the reference is
 * undefined or broken: errors return -1 instead of
 * crashing, the decoder never reads outside a waveform's own words, and
 * M = 1 is given its natural meaning (escape iff z >= 8).
 */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#if defined(_OPENMP)
#include <omp.h>
#endif

#define DRO_MAX_TAPS 64
#define DRO_ESCAPE_Q 8u /* "giveup", src/deltaRice.c:203 */

typedef struct {
    int k;            /* log2(M), 0..15 */
    long wave_len;    /* samples per waveform; <=0 means "whole chunk" */
    int n_taps;
    int taps[DRO_MAX_TAPS];
    int is_delta;     /* taps == [1,-1]  (src/deltaRice.c:38-46) */
} dro_opts;

/* ---- option parsing: src/deltaRice.c:248-291 + :114-136 ------------------ */

/* log2 of a power of two in [1, 32768]; -1 otherwise (src/deltaRice.c:114-136;
 * the upper limit is SURVEY.md Appendix B4: M >= 65536 corrupts in the reference). */
int dro_log2_m(long m) {
    if (m <= 0 || (m & (m - 1)) != 0 || m > 32768) return -1;
    int k = 0;
    while ((1L << k) != m) ++k;
    return k;
}

int dro_parse_opts(const unsigned *cd, size_t n, dro_opts *o) {
    long m = 8;
    o->wave_len = -1;
    o->n_taps = 2;
    o->taps[0] = 1;
    o->taps[1] = -1;
    if (n >= 1) m = (long)(int)cd[0];
    if (n >= 2) o->wave_len = (long)(int)cd[1];
    if (n >= 3) {
        long nt = (long)(int)cd[2];
        if (nt <= 0 || nt > DRO_MAX_TAPS || (size_t)nt + 3 > n) return -1;
        o->n_taps = (int)nt;
        for (long j = 0; j < nt; ++j) o->taps[j] = (int)cd[3 + j];
        if (o->taps[0] == 0) return -1; /* IIR divides by taps[0], :99 */
    }
    o->k = dro_log2_m(m);
    if (o->k < 0) return -1;
    if (o->wave_len == 0 || o->wave_len < -1) return -1;
    o->is_delta = (o->n_taps == 2 && o->taps[0] == 1 && o->taps[1] == -1);
    return 0;
}

/* ---- prediction filter --------------------------------------------------- */

/* Forward filter: d[i] = sum_j taps[j]*x[i-j], every partial sum truncated to
 * int16 (src/deltaRice.c:65-73); truncation commutes with addition mod 2^16,
 * so one wrap at the end is the same value.  Delta special case :51-63. */
void dro_filter_forward(const int16_t *x, int16_t *d, long n, const dro_opts *o) {
    for (long i = 0; i < n; ++i) {
        uint32_t acc = 0;
        for (int j = 0; j < o->n_taps && j <= i; ++j)
            acc += (uint32_t)((int32_t)x[i - j] * o->taps[j]);
        d[i] = (int16_t)(uint16_t)acc;
    }
}

/* Inverse: y[i] = (int16)( (int16)(d[i] - sum_{j>=1} taps[j]*y[i-j]) / taps[0] )
 * (src/deltaRice.c:92-101); delta special case = running sum (:80-89). */
void dro_filter_inverse(const int16_t *d, int16_t *y, long n, const dro_opts *o) {
    for (long i = 0; i < n; ++i) {
        uint32_t acc = (uint32_t)(int32_t)d[i];
        for (int j = 1; j < o->n_taps && j <= i; ++j)
            acc -= (uint32_t)((int32_t)y[i - j] * o->taps[j]);
        int32_t t = (int16_t)(uint16_t)acc;
        y[i] = (int16_t)(t / o->taps[0]);
    }
}

/* ---- Rice coding of one waveform ----------------------------------------- */

static inline uint32_t zigzag16(int16_t d) { /* src/deltaRice.c:207-211 */
    int32_t v = d;
    return (uint32_t)(v >= 0 ? 2 * v : -2 * v - 1); /* 0..65535 */
}

static inline int16_t unzigzag16(uint32_t z) { /* src/deltaRice.c:172-177 */
    return (int16_t)((z & 1u) ? -(int32_t)((z + 1u) >> 1) : (int32_t)(z >> 1));
}

/* Number of code bits for residual d (src/deltaRice.c:215-228). */
static inline unsigned code_bits(int16_t d, int k) {
    uint32_t z = zigzag16(d);
    uint32_t q = z >> k;
    return q < DRO_ESCAPE_Q ? q + 1u + (unsigned)k : DRO_ESCAPE_Q + 1u + 16u;
}

/* Packs n residuals; returns the number of u32 words written (<= n for n>0).
 * out must hold dro_max_wave_words(n) words. */
long dro_rice_pack(const int16_t *d, long n, int k, uint32_t *out) {
    uint64_t acc = 0; /* pending bits, right-aligned */
    unsigned pend = 0;
    long nw = 0;
    for (long i = 0; i < n; ++i) {
        uint32_t z = zigzag16(d[i]);
        uint32_t q = z >> k;
        if (q < DRO_ESCAPE_Q) {
            acc = (acc << (q + 1)) | 1u;
            acc = (acc << k) | (z & ((1u << k) - 1u));
            pend += q + 1u + (unsigned)k;
        } else {
            acc = (acc << (DRO_ESCAPE_Q + 1)) | 1u;
            acc = (acc << 16) | z;
            pend += DRO_ESCAPE_Q + 1u + 16u;
        }
        if (pend >= 32) { /* at most one word per sample: code <= 25 bits */
            pend -= 32;
            out[nw++] = (uint32_t)(acc >> pend);
            acc &= (((uint64_t)1) << pend) - 1u;
        }
    }
    if (pend) out[nw++] = (uint32_t)(acc << (32 - pend));
    return nw;
}

/* Reads bit `pos` (MSB-first) of a word array of `nw` words; 0 past the end. */
static inline unsigned get_bit(const uint32_t *w, long nw, uint64_t pos) {
    uint64_t wi = pos >> 5;
    if ((long)wi >= nw) return 0;
    return (w[wi] >> (31u - (unsigned)(pos & 31u))) & 1u;
}

static inline uint32_t get_bits(const uint32_t *w, long nw, uint64_t pos, unsigned n) {
    uint32_t v = 0;
    for (unsigned b = 0; b < n; ++b) v = (v << 1) | get_bit(w, nw, pos + b);
    return v;
}

/* Unpacks n residuals from nw words (src/deltaRice.c:138-189).  Returns the
 * number of bits consumed, or -1 if the stream runs out / has >8 leading zeros
 * (the reference does not check; it reads on). */
long dro_rice_unpack(const uint32_t *in, long nw, long n, int k, int16_t *d) {
    uint64_t pos = 0;
    const uint64_t end = (uint64_t)nw * 32u;
    for (long i = 0; i < n; ++i) {
        unsigned q = 0;
        while (pos < end && !get_bit(in, nw, pos)) {
            ++q;
            ++pos;
            if (q > DRO_ESCAPE_Q) return -1;
        }
        if (pos >= end) return -1;
        ++pos; /* the terminating 1 */
        uint32_t z;
        if (q == DRO_ESCAPE_Q) {
            z = get_bits(in, nw, pos, 16);
            pos += 16;
        } else {
            z = (q << k) + get_bits(in, nw, pos, (unsigned)k);
            pos += (unsigned)k;
        }
        if (pos > end) return -1;
        d[i] = unzigzag16(z);
    }
    return (long)pos;
}

/* Word-at-a-time decoder with the same results; used for the timed CPU
 * baseline so that the "port" is not handicapped by the bit loop above. */
static long rice_unpack_fast(const uint32_t *in, long nw, long n, int k, int16_t *d) {
    uint64_t win = 0; /* next bits, left-aligned */
    unsigned have = 0;
    long wi = 0;
    uint64_t used = 0;
    for (long i = 0; i < n; ++i) {
        while (have <= 32 && wi < nw) {
            win |= (uint64_t)in[wi++] << (32 - have);
            have += 32;
        }
        unsigned q = win ? (unsigned)__builtin_clzll(win) : 64u;
        if (q > DRO_ESCAPE_Q) return -1;
        unsigned pl = (q == DRO_ESCAPE_Q) ? 16u : (unsigned)k;
        unsigned len = q + 1u + pl;
        if (len > have) return -1;
        uint64_t t = win << (q + 1u);
        uint32_t r = pl ? (uint32_t)(t >> (64u - pl)) : 0u;
        uint32_t z = (q == DRO_ESCAPE_Q) ? r : ((q << k) + r);
        d[i] = unzigzag16(z);
        win <<= len;
        have -= len;
        used += len;
    }
    return (long)used;
}

/* ---- chunk framing -------------------------------------------------------- */

/* Upper bound on payload words of one waveform of n samples: 25 bits/sample. */
size_t dro_max_wave_words(size_t n) { return (n * 25u + 31u) / 32u; }

static long chunk_geometry(long total, const dro_opts *o, long *L, long *last_len) {
    long wl = o->wave_len <= 0 ? total : o->wave_len; /* :391-393, :307-308 */
    long nw = total / wl;
    long left = total - nw * wl; /* :399-403 */
    if (left) ++nw;
    *L = wl;
    *last_len = left ? left : wl;
    return nw;
}

/* Upper bound on words of an encoded chunk of n samples with waveform length L (<=0: whole). */
size_t dro_max_chunk_words(size_t n, long L) {
    size_t wl = (L <= 0) ? n : (size_t)L;
    if (wl == 0) return 1;
    size_t nw = (n + wl - 1) / wl;
    return 1 + nw + dro_max_wave_words(n) + nw; /* per-wave rounding slack */
}

/* Encodes one chunk (src/deltaRice.c:383-441, OpenMP branch).  Returns words
 * written, -1 on invalid arguments, -2 if out_cap is too small. */
long dro_encode_chunk(const int16_t *in, size_t nbytes, const unsigned *cd, size_t cd_n,
                      uint32_t *out, size_t out_cap) {
    dro_opts o;
    if (dro_parse_opts(cd, cd_n, &o) != 0) return -1;
    if (nbytes == 0 || (nbytes & 1u) || nbytes / 2 > 0x7fffffffUL) return -1;
    const long total = (long)(nbytes / 2);
    long L, last_len;
    const long W = chunk_geometry(total, &o, &L, &last_len);
    if (out_cap < dro_max_chunk_words((size_t)total, L)) return -2;

    /* staging: waveform i at word i*(L+1)+1, as the reference does (:421-424) */
    uint32_t *stage = (uint32_t *)malloc(sizeof(uint32_t) * ((size_t)W * (size_t)(L + 1) + 1));
    int16_t *resid = (int16_t *)malloc(sizeof(int16_t) * (size_t)total);
    long *nwords = (long *)malloc(sizeof(long) * (size_t)W);
    if (!stage || !resid || !nwords) {
        free(stage); free(resid); free(nwords);
        return -1;
    }
    long i;
#pragma omp parallel for schedule(static)
    for (i = 0; i < W; ++i) {
        const long len = (i == W - 1) ? last_len : L;
        dro_filter_forward(in + i * L, resid + i * L, len, &o);
        nwords[i] = dro_rice_pack(resid + i * L, len, o.k, stage + (size_t)i * (size_t)(L + 1) + 1);
    }
    size_t at = 0;
    out[at++] = (uint32_t)total; /* :415 */
    for (i = 0; i < W; ++i) {
        out[at++] = (uint32_t)nwords[i]; /* :379 */
        memcpy(out + at, stage + (size_t)i * (size_t)(L + 1) + 1, sizeof(uint32_t) * (size_t)nwords[i]);
        at += (size_t)nwords[i];
    }
    free(stage); free(resid); free(nwords);
    return (long)at;
}

/* Decodes one chunk (src/deltaRice.c:301-341).  Returns samples written, -1 on
 * invalid arguments or a corrupt stream, -2 if out_cap is too small. */
static long decode_chunk_impl(const uint32_t *in, size_t nbytes, const unsigned *cd, size_t cd_n,
                              int16_t *out, size_t out_cap, int fast) {
    dro_opts o;
    if (dro_parse_opts(cd, cd_n, &o) != 0) return -1;
    if (nbytes < 8 || (nbytes & 3u)) return -1;
    const long nwords_in = (long)(nbytes / 4);
    if (in[0] == 0 || in[0] > 0x7fffffffU) return -1;
    const long total = (long)in[0]; /* :306 */
    if ((size_t)total > out_cap) return -2;
    long L, last_len;
    const long W = chunk_geometry(total, &o, &L, &last_len);

    long *start = (long *)malloc(sizeof(long) * (size_t)(W + 1));
    int16_t *resid = (int16_t *)malloc(sizeof(int16_t) * (size_t)total);
    if (!start || !resid) { free(start); free(resid); return -1; }
    long at = 1; /* header chain walk, :320-325 */
    int bad = 0;
    for (long w = 0; w < W; ++w) {
        if (at >= nwords_in) { bad = 1; break; }
        start[w] = at;
        at += (long)in[at] + 1;
    }
    if (bad || at != nwords_in) { free(start); free(resid); return -1; }
    start[W] = at;

    long i;
#pragma omp parallel for schedule(static) reduction(| : bad)
    for (i = 0; i < W; ++i) {
        const long len = (i == W - 1) ? last_len : L;
        const long n_i = (long)in[start[i]];
        long used = fast ? rice_unpack_fast(in + start[i] + 1, n_i, len, o.k, resid + i * L)
                         : dro_rice_unpack(in + start[i] + 1, n_i, len, o.k, resid + i * L);
        if (used < 0) { bad |= 1; continue; }
        dro_filter_inverse(resid + i * L, out + i * L, len, &o);
    }
    free(start); free(resid);
    return bad ? -1 : total;
}

long dro_decode_chunk(const uint32_t *in, size_t nbytes, const unsigned *cd, size_t cd_n,
                      int16_t *out, size_t out_cap) {
    return decode_chunk_impl(in, nbytes, cd, cd_n, out, out_cap, 0);
}

long dro_decode_chunk_fast(const uint32_t *in, size_t nbytes, const unsigned *cd, size_t cd_n,
                           int16_t *out, size_t out_cap) {
    return decode_chunk_impl(in, nbytes, cd, cd_n, out, out_cap, 1);
}

/* ---- batch helpers (uniform chunks), used by tests and the CPU baseline ---- */

/* Encodes n_chunks chunks of chunk_samples int16 each, back to back, into out;
 * chunk_word_off[c] = first word of chunk c, chunk_word_off[n_chunks] = total. */
long dro_encode_batch(const int16_t *in, size_t n_chunks, size_t chunk_samples,
                      const unsigned *cd, size_t cd_n, uint32_t *out, size_t out_cap,
                      uint64_t *chunk_word_off) {
    size_t at = 0;
    for (size_t c = 0; c < n_chunks; ++c) {
        chunk_word_off[c] = at;
        long w = dro_encode_chunk(in + c * chunk_samples, chunk_samples * 2, cd, cd_n,
                                  out + at, out_cap - at);
        if (w < 0) return w;
        at += (size_t)w;
    }
    chunk_word_off[n_chunks] = at;
    return (long)at;
}

long dro_decode_batch(const uint32_t *in, size_t n_chunks, const uint64_t *chunk_word_off,
                      size_t chunk_samples, const unsigned *cd, size_t cd_n, int16_t *out) {
    for (size_t c = 0; c < n_chunks; ++c) {
        size_t nb = (size_t)(chunk_word_off[c + 1] - chunk_word_off[c]) * 4u;
        long r = decode_chunk_impl(in + chunk_word_off[c], nb, cd, cd_n,
                                   out + c * chunk_samples, chunk_samples, 1);
        if (r != (long)chunk_samples) return -1;
    }
    return (long)(n_chunks * chunk_samples);
}

int dro_num_threads(void) {
#if defined(_OPENMP)
    return omp_get_max_threads();
#else
    return 1;
#endif
}
