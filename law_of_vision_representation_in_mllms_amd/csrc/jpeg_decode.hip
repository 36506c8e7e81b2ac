// JPEG decode for the device input pipeline: replaces PIL's Image.open(...).convert('RGB') of the reference's loaders
// (C_score/extract_feature.py:65-66, llava/mm_utils.py:78-95, llava/feature/extract.py) bit for bit.
//
// Split the way the work wants it:
//   host  (visrep_jpeg_info, visrep_jpeg_entropy_decode): marker parsing and baseline Huffman decoding - a serial bit stream, one image
//         per host thread - into QUANTISED coefficients (int16, natural 8x8 order, component planes of whole MCUs) + the quantisation tables;
//   device (visrep_jpeg_reconstruct): dequantisation + the accurate-integer inverse DCT (libjpeg's "islow", the default dct_method),
//         "fancy" (triangle-filter) chroma upsampling for 4:2:0 / 4:2:2, YCbCr -> RGB with libjpeg's 16-bit fixed-point tables - one
//         launch pair for a whole batch of ragged images, output packed RGB u8 [H, W, 3] per image (what device_preprocess.resize_u8 takes).
// What PIL runs underneath is libjpeg-turbo (Pillow 12.2 bundles 3.x, API level 6.2): jidctint.c:jpeg_idct_islow, jdsample.c:
// h2v1_fancy_upsample / h2v2_fancy_upsample (+ the context-row replication of jdmainct.c), jdcolor.c:ycc_rgb_convert.  Those are a
// third-party dependency of the reference (absent from /root/reference); their arithmetic is restated here and in oracle/jpeg.py, and
// pinned against PIL itself (tests/test_host_jpeg.py on the CPU, tests/test_gpu_jpeg.py on the device).
// Not decodable here (visrep_jpeg_info reports why; the Python side then lets PIL decode that file and counts it): progressive and
// arithmetic-coded files, 12-bit samples, CMYK / YCCK, non-interleaved multi-scan files, chroma layouts other than 4:4:4 / 4:2:2 / 4:2:0.
#include <cstdint>
#include <cstring>

#include "common.h"
#include "visrep_internal.h"

namespace {

const uint8_t kNatural[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                              41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                              30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};   // zigzag position -> row-major index

struct Huff {
    uint8_t bits[17];
    uint8_t vals[256];
    int maxcode[18];       // largest code of each length (-1 if none), maxcode[17] = sentinel
    int valoff[17];
    uint16_t look[512];    // 9-bit lookahead: (length << 8) | symbol, 0 = longer than 9 bits
    bool present;
};

struct Parsed {
    VisrepJpegInfo info;
    uint16_t qt[4][64];    // natural order
    bool qt_ok[4];
    Huff dc[4], ac[4];
    int comp_id[3], comp_tq[3], comp_td[3], comp_ta[3];
    size_t scan_off;       // first entropy-coded byte
};

bool build_huff(Huff& h) {
    int code = 0, k = 0;
    uint16_t codes[256];
    uint8_t sizes[256];
    for (int l = 1; l <= 16; ++l) {
        if (k + h.bits[l] > 256) return false;
        for (int i = 0; i < h.bits[l]; ++i) { sizes[k] = (uint8_t)l; codes[k] = (uint16_t)code; ++code; ++k; }
        if (code > (1 << l)) return false;
        code <<= 1;
    }
    int p = 0;
    for (int l = 1; l <= 16; ++l) {
        if (h.bits[l]) {
            h.valoff[l] = p - (int)codes[p];
            p += h.bits[l];
            h.maxcode[l] = codes[p - 1];
        } else {
            h.maxcode[l] = -1;
        }
    }
    h.maxcode[17] = 0xFFFFF;
    memset(h.look, 0, sizeof(h.look));
    for (int i = 0; i < k; ++i)
        if (sizes[i] <= 9) {
            const int base = codes[i] << (9 - sizes[i]);
            for (int j = 0; j < (1 << (9 - sizes[i])); ++j) h.look[base + j] = (uint16_t)((sizes[i] << 8) | h.vals[i]);
        }
    h.present = true;
    return true;
}

inline int rd16(const uint8_t* p) { return (p[0] << 8) | p[1]; }

// Marker segments up to the first SOS.  Returns 0, or an error code with `why` set.
int parse(const uint8_t* d, size_t n, Parsed& P, const char*& why) {
    memset(&P, 0, sizeof(P));
    VisrepJpegInfo& I = P.info;
    if (n < 4 || d[0] != 0xFF || d[1] != 0xD8) { why = "not a JPEG stream (no SOI)"; return VISREP_ERR_ARG; }
    size_t pos = 2;
    bool sof = false;
    int adobe_transform = -1;
    while (pos + 4 <= n) {
        if (d[pos] != 0xFF) { why = "marker expected"; return VISREP_ERR_ARG; }
        while (pos < n && d[pos] == 0xFF) ++pos;                   // fill bytes
        if (pos >= n) break;
        const int m = d[pos++];
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        if (pos + 2 > n) break;
        const int len = rd16(d + pos);
        if (len < 2 || pos + len > n) { why = "truncated marker segment"; return VISREP_ERR_ARG; }
        const uint8_t* s = d + pos + 2;
        const int sl = len - 2;
        if (m == 0xDB) {                                           // DQT
            int o = 0;
            while (o < sl) {
                const int pq = s[o] >> 4, tq = s[o] & 15;
                ++o;
                if (tq > 3 || pq > 1 || o + 64 * (pq + 1) > sl) { why = "bad DQT"; return VISREP_ERR_ARG; }
                for (int k = 0; k < 64; ++k) { P.qt[tq][kNatural[k]] = (uint16_t)(pq ? rd16(s + o + 2 * k) : s[o + k]); }
                o += 64 * (pq + 1);
                P.qt_ok[tq] = true;
            }
        } else if (m == 0xC4) {                                    // DHT
            int o = 0;
            while (o < sl) {
                if (o + 17 > sl) { why = "bad DHT"; return VISREP_ERR_ARG; }
                const int tc = s[o] >> 4, th = s[o] & 15;
                if (tc > 1 || th > 3) { why = "bad DHT"; return VISREP_ERR_ARG; }
                Huff& h = tc ? P.ac[th] : P.dc[th];
                int cnt = 0;
                h.bits[0] = 0;
                for (int l = 1; l <= 16; ++l) { h.bits[l] = s[o + l]; cnt += h.bits[l]; }
                o += 17;
                if (cnt > 256 || o + cnt > sl) { why = "bad DHT"; return VISREP_ERR_ARG; }
                memcpy(h.vals, s + o, cnt);
                o += cnt;
                if (!build_huff(h)) { why = "bad Huffman table"; return VISREP_ERR_ARG; }
            }
        } else if (m == 0xC0 || m == 0xC1 || m == 0xC2) {          // SOF0 / SOF1 / SOF2
            if (sof) { why = "two frame headers"; return VISREP_ERR_ARG; }
            sof = true;
            if (sl < 6) { why = "bad SOF"; return VISREP_ERR_ARG; }
            I.progressive = m == 0xC2;
            const int prec = s[0];
            I.height = rd16(s + 1);
            I.width = rd16(s + 3);
            I.ncomp = s[5];
            if (prec != 8) { why = "12-bit samples"; I.unsupported = 1; }
            if (I.ncomp != 1 && I.ncomp != 3) { why = "CMYK / YCCK (4 components)"; I.unsupported = 1; if (I.ncomp > 3 || I.ncomp < 1) return VISREP_ERR_SHAPE; }
            if (sl < 6 + 3 * I.ncomp || I.width <= 0 || I.height <= 0) { why = "bad SOF"; return VISREP_ERR_ARG; }
            for (int c = 0; c < I.ncomp; ++c) {
                P.comp_id[c] = s[6 + 3 * c];
                I.hs[c] = s[7 + 3 * c] >> 4;
                I.vs[c] = s[7 + 3 * c] & 15;
                P.comp_tq[c] = s[8 + 3 * c] & 3;
                if (I.hs[c] < 1 || I.hs[c] > 4 || I.vs[c] < 1 || I.vs[c] > 4) { why = "bad sampling factors"; return VISREP_ERR_ARG; }
            }
        } else if (m == 0xC9 || m == 0xCA || m == 0xCB || (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC)) {
            why = "arithmetic-coded, lossless or hierarchical JPEG";
            I.unsupported = 1;
            return VISREP_ERR_SHAPE;
        } else if (m == 0xDD) {                                    // DRI
            if (sl >= 2) I.restart_interval = rd16(s);
        } else if (m == 0xEE) {                                    // APP14 "Adobe": colour transform flag
            if (sl >= 12 && !memcmp(s, "Adobe", 5)) adobe_transform = s[11];
        } else if (m == 0xDA) {                                    // SOS
            if (!sof) { why = "scan before frame header"; return VISREP_ERR_ARG; }
            if (sl < 1) { why = "bad SOS"; return VISREP_ERR_ARG; }
            const int ns = s[0];
            if (sl < 1 + 2 * ns + 3) { why = "bad SOS"; return VISREP_ERR_ARG; }
            if (ns != I.ncomp) { why = "non-interleaved (multi-scan) file"; I.unsupported = 1; }
            for (int j = 0; j < ns && j < 3; ++j) {
                int c = -1;
                for (int q = 0; q < I.ncomp; ++q)
                    if (P.comp_id[q] == s[1 + 2 * j]) c = q;
                if (c != j) { why = "scan components out of frame order"; I.unsupported = 1; c = j; }
                P.comp_td[c] = s[2 + 2 * j] >> 4;
                P.comp_ta[c] = s[2 + 2 * j] & 15;
                if (P.comp_td[c] > 3 || P.comp_ta[c] > 3) { why = "bad SOS"; return VISREP_ERR_ARG; }
            }
            P.scan_off = pos + len;
            break;
        }
        pos += len;
    }
    if (!sof || !P.scan_off) { why = "no frame / scan found"; return VISREP_ERR_ARG; }
    // geometry (jdmaster.c / jdinput.c: per-component sizes)
    I.hmax = I.vmax = 1;
    for (int c = 0; c < I.ncomp; ++c) { I.hmax = I.hs[c] > I.hmax ? I.hs[c] : I.hmax; I.vmax = I.vs[c] > I.vmax ? I.vs[c] : I.vmax; }
    if (I.ncomp == 1) { I.hs[0] = I.vs[0] = I.hmax = I.vmax = 1; }     // a single-component scan is never interleaved: MCU = one block
    I.mcus_w = (I.width + 8 * I.hmax - 1) / (8 * I.hmax);
    I.mcus_h = (I.height + 8 * I.vmax - 1) / (8 * I.vmax);
    I.coef_count = 0;
    for (int c = 0; c < I.ncomp; ++c) {
        I.blocks_w[c] = I.mcus_w * I.hs[c];
        I.blocks_h[c] = I.mcus_h * I.vs[c];
        I.comp_w[c] = (I.width * I.hs[c] + I.hmax - 1) / I.hmax;
        I.comp_h[c] = (I.height * I.vs[c] + I.vmax - 1) / I.vmax;
        I.coef_count += (long)I.blocks_w[c] * I.blocks_h[c] * 64;
    }
    if (I.progressive) { why = "progressive JPEG"; I.unsupported = 1; }
    if (I.ncomp == 3) {
        if (adobe_transform == 0) { why = "Adobe RGB (untransformed) JPEG"; I.unsupported = 1; }
        if (P.comp_id[0] == 'R' && P.comp_id[1] == 'G' && P.comp_id[2] == 'B' && adobe_transform < 0) { why = "RGB component ids"; I.unsupported = 1; }
        const bool luma_full = I.hs[0] == I.hmax && I.vs[0] == I.vmax;
        const bool chroma_ok = I.hs[1] == 1 && I.vs[1] == 1 && I.hs[2] == 1 && I.vs[2] == 1 &&
                               ((I.hmax == 1 && I.vmax == 1) || (I.hmax == 2 && I.vmax == 1) || (I.hmax == 2 && I.vmax == 2));
        if (!luma_full || !chroma_ok) { why = "chroma layout other than 4:4:4 / 4:2:2 / 4:2:0"; I.unsupported = 1; }
    }
    for (int c = 0; c < I.ncomp; ++c)
        if (!P.qt_ok[P.comp_tq[c]]) { why = "missing quantisation table"; return VISREP_ERR_ARG; }
    return 0;
}

// ---- bit reader over the entropy-coded segment (byte stuffing, restart markers)
struct Bits {
    const uint8_t* d; size_t n, pos;
    uint64_t acc; int cnt;                 // acc holds cnt valid bits, msb first
    bool hit_marker;
    int fake;                              // zero bytes fed past a marker / the end of the data (the newest `fake` bytes of acc)
    // True once the decoder has CONSUMED bits that were not in the file: the entropy-coded segment ended (a marker, or the data itself)
    // before the MCUs it should hold.  libjpeg pads such a scan with zeros and warns; PIL raises "image file is truncated" when the data
    // ends - either way the reference does not silently produce these blocks, so the caller falls back to PIL (which raises like the reference).
    inline bool overran() const { return cnt < 8 * fake; }
    void fill() {
        while (cnt <= 48) {
            int b = 0;
            const bool was = hit_marker;
            if (!hit_marker && pos < n) {
                b = d[pos];
                if (b == 0xFF) {
                    const int b2 = pos + 1 < n ? d[pos + 1] : 0xD9;
                    if (b2 == 0) pos += 2;                 // stuffed zero
                    else { hit_marker = true; b = 0; }     // a marker: feed zeros (jdhuff.c does the same past the end of the data)
                } else {
                    ++pos;
                }
            } else if (pos >= n) {
                hit_marker = true;
            }
            if (hit_marker || was) ++fake;
            acc = (acc << 8) | (uint64_t)b;
            cnt += 8;
        }
    }
    inline int peek(int nb) { if (cnt < nb) fill(); return (int)((acc >> (cnt - nb)) & ((1u << nb) - 1)); }
    inline void skip(int nb) { cnt -= nb; }
    inline int get(int nb) { if (!nb) return 0; const int v = peek(nb); skip(nb); return v; }
};

inline int extend(int v, int t) { return v < (1 << (t - 1)) ? v - (1 << t) + 1 : v; }     // HUFF_EXTEND

inline int decode_sym(Bits& b, const Huff& h) {
    const int look = b.peek(9);
    const uint16_t e = h.look[look];
    if (e) { b.skip(e >> 8); return e & 255; }
    int code = look, l = 9;
    // slow path (codes longer than 9 bits): jdhuff.c jpeg_huff_decode
    b.skip(9);
    for (;;) {
        if (l > 16) return -1;
        if (h.maxcode[l] >= 0 && code <= h.maxcode[l]) break;
        code = (code << 1) | b.get(1);
        ++l;
    }
    // l == 9 with no lookup entry means the 9-bit prefix belongs to a longer code: handled by the loop (it only breaks on a full code)
    const int idx = code + h.valoff[l];
    return idx >= 0 && idx < 256 ? h.vals[idx] : -1;
}

}  // namespace

extern "C" int visrep_jpeg_info(const void* data, size_t n, VisrepJpegInfo* info) {
    if (!data || !info) return visrep_set_error(VISREP_ERR_ARG, "jpeg_info: null pointer");
    static thread_local Parsed P;
    const char* why = "";
    const int rc = parse((const uint8_t*)data, n, P, why);
    *info = P.info;
    if (rc) return visrep_set_error(rc, why);
    if (P.info.unsupported) visrep_set_error(VISREP_ERR_SHAPE, why);          // rc stays 0: the caller reads info->unsupported and the message
    return 0;
}

extern "C" int visrep_jpeg_entropy_decode(const void* data, size_t n, int16_t* coef, uint16_t* qtab) {
    if (!data || !coef || !qtab) return visrep_set_error(VISREP_ERR_ARG, "jpeg_entropy_decode: null pointer");
    static thread_local Parsed P;
    const char* why = "";
    const int rc = parse((const uint8_t*)data, n, P, why);
    if (rc) return visrep_set_error(rc, why);
    const VisrepJpegInfo& I = P.info;
    if (I.unsupported) return visrep_set_error(VISREP_ERR_SHAPE, why);
    for (int c = 0; c < I.ncomp; ++c) {
        if (!P.dc[P.comp_td[c]].present || !P.ac[P.comp_ta[c]].present) return visrep_set_error(VISREP_ERR_ARG, "jpeg: scan uses an undefined Huffman table");
        memcpy(qtab + 64 * c, P.qt[P.comp_tq[c]], 128);
    }
    memset(coef, 0, (size_t)I.coef_count * sizeof(int16_t));
    long plane[3];
    plane[0] = 0;
    for (int c = 1; c < I.ncomp; ++c) plane[c] = plane[c - 1] + (long)I.blocks_w[c - 1] * I.blocks_h[c - 1] * 64;
    Bits b{(const uint8_t*)data, n, P.scan_off, 0, 0, false, 0};
    int pred[3] = {0, 0, 0};
    const long nmcu = (long)I.mcus_w * I.mcus_h;
    int to_restart = I.restart_interval;
    for (long m = 0; m < nmcu; ++m) {
        if (I.restart_interval && to_restart == 0) {
            // byte-align, expect RSTn, reset the predictors (jdhuff.c process_restart)
            if (b.overran()) return visrep_set_error(VISREP_ERR_ARG, "jpeg: entropy-coded data ends before its restart interval (truncated or corrupt)");
            b.acc = 0; b.cnt = 0; b.hit_marker = false; b.fake = 0;
            while (b.pos + 1 < n && !(b.d[b.pos] == 0xFF && b.d[b.pos + 1] >= 0xD0 && b.d[b.pos + 1] <= 0xD7)) ++b.pos;   // skip to the marker
            if (b.pos + 1 >= n) return visrep_set_error(VISREP_ERR_ARG, "jpeg: restart marker missing (truncated or corrupt)");
            b.pos += 2;
            pred[0] = pred[1] = pred[2] = 0;
            to_restart = I.restart_interval;
        }
        const int my = (int)(m / I.mcus_w), mx = (int)(m % I.mcus_w);
        for (int c = 0; c < I.ncomp; ++c) {
            const Huff& hd = P.dc[P.comp_td[c]];
            const Huff& ha = P.ac[P.comp_ta[c]];
            for (int v = 0; v < I.vs[c]; ++v)
                for (int h = 0; h < I.hs[c]; ++h) {
                    int16_t* blk = coef + plane[c] + ((long)(my * I.vs[c] + v) * I.blocks_w[c] + (mx * I.hs[c] + h)) * 64;
                    int s = decode_sym(b, hd);
                    if (s < 0 || s > 15) return visrep_set_error(VISREP_ERR_ARG, "jpeg: corrupt DC code");
                    int diff = 0;
                    if (s) diff = extend(b.get(s), s);
                    pred[c] += diff;
                    blk[0] = (int16_t)pred[c];
                    for (int k = 1; k < 64;) {
                        const int rs = decode_sym(b, ha);
                        if (rs < 0) return visrep_set_error(VISREP_ERR_ARG, "jpeg: corrupt AC code");
                        const int r = rs >> 4, sz = rs & 15;
                        if (sz) {
                            k += r;
                            if (k > 63) return visrep_set_error(VISREP_ERR_ARG, "jpeg: AC run past the block");
                            blk[kNatural[k]] = (int16_t)extend(b.get(sz), sz);
                            ++k;
                        } else {
                            if (r != 15) break;                    // EOB
                            k += 16;
                        }
                    }
                }
        }
        if (I.restart_interval) --to_restart;
    }
    if (b.overran()) return visrep_set_error(VISREP_ERR_ARG, "jpeg: entropy-coded data ends before the last MCU (truncated or corrupt)");
    return 0;
}

// ------------------------------------------------------------------------------------------------ device side
namespace {

// descriptor of one image (long fields; built by device_jpeg.py): see JD_* below
enum { JD_COEF0 = 0, JD_COEF1, JD_COEF2, JD_PLANE0, JD_PLANE1, JD_PLANE2, JD_BW0, JD_BW1, JD_BW2, JD_BH0, JD_BH1, JD_BH2, JD_CW0, JD_CW1, JD_CW2,
       JD_CH0, JD_CH1, JD_CH2, JD_W, JD_H, JD_NCOMP, JD_HMAX, JD_VMAX, JD_RGB, JD_QT, JD_FIELDS = 32 };

VR_DEV int range_limit_idct(int x) {          // jdmaster.c prepare_range_limit_table, the post-IDCT half: index x & 1023
    const int i = x & 1023;
    return i < 128 ? i + 128 : i < 512 ? 255 : i < 896 ? 0 : i - 896;
}

// jidctint.c jpeg_idct_islow (CONST_BITS 13, PASS1_BITS 2): one thread = one 8x8 block, 32-bit wrap-around arithmetic like the SIMD
// implementations libjpeg-turbo runs (their inputs are 16-bit products, every accumulation is 32-bit)
VR_DEV void idct_1d(int i0, int i1, int i2, int i3, int i4, int i5, int i6, int i7, int shift, int* o) {
    unsigned z2 = (unsigned)i2, z3 = (unsigned)i6;
    unsigned z1 = (z2 + z3) * 4433u;
    unsigned tmp2 = z1 + z3 * (unsigned)(-15137);
    unsigned tmp3 = z1 + z2 * 6270u;
    z2 = (unsigned)i0; z3 = (unsigned)i4;
    unsigned tmp0 = (z2 + z3) << 13, tmp1 = (z2 - z3) << 13;
    const unsigned tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = (unsigned)i7; tmp1 = (unsigned)i5; tmp2 = (unsigned)i3; tmp3 = (unsigned)i1;
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
    unsigned z4 = tmp1 + tmp3;
    const unsigned z5 = (z3 + z4) * 9633u;
    tmp0 *= 2446u; tmp1 *= 16819u; tmp2 *= 25172u; tmp3 *= 12299u;
    z1 *= (unsigned)(-7373); z2 *= (unsigned)(-20995); z3 *= (unsigned)(-16069); z4 *= (unsigned)(-3196);
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    const unsigned rnd = 1u << (shift - 1);
    o[0] = (int)(tmp10 + tmp3 + rnd) >> shift; o[7] = (int)(tmp10 - tmp3 + rnd) >> shift;
    o[1] = (int)(tmp11 + tmp2 + rnd) >> shift; o[6] = (int)(tmp11 - tmp2 + rnd) >> shift;
    o[2] = (int)(tmp12 + tmp1 + rnd) >> shift; o[5] = (int)(tmp12 - tmp1 + rnd) >> shift;
    o[3] = (int)(tmp13 + tmp0 + rnd) >> shift; o[4] = (int)(tmp13 - tmp0 + rnd) >> shift;
}

__global__ __launch_bounds__(64) void jpeg_idct_kernel(const int16_t* __restrict__ coef, const uint16_t* __restrict__ qtab, const long* __restrict__ desc,
                                                      uint8_t* __restrict__ planes) {
    const long* D = desc + (long)blockIdx.y * JD_FIELDS;
    long blk = (long)blockIdx.x * 64 + threadIdx.x;
    int c = 0;
    const int nc = (int)D[JD_NCOMP];
    for (; c < nc; ++c) {
        const long nb = D[JD_BW0 + c] * D[JD_BH0 + c];
        if (blk < nb) break;
        blk -= nb;
    }
    if (c == nc) return;
    const int bw = (int)D[JD_BW0 + c];
    const int by = (int)(blk / bw), bx = (int)(blk % bw);
    const int16_t* in = coef + D[JD_COEF0 + c] + blk * 64;
    const uint16_t* q = qtab + D[JD_QT] + 64 * c;
    int ws[64];
#pragma unroll
    for (int col = 0; col < 8; ++col) {
        int o[8];
        idct_1d((int)in[col] * q[col], (int)in[8 + col] * q[8 + col], (int)in[16 + col] * q[16 + col], (int)in[24 + col] * q[24 + col],
                (int)in[32 + col] * q[32 + col], (int)in[40 + col] * q[40 + col], (int)in[48 + col] * q[48 + col], (int)in[56 + col] * q[56 + col], 11, o);
#pragma unroll
        for (int r = 0; r < 8; ++r) ws[8 * r + col] = o[r];
    }
    uint8_t* out = planes + D[JD_PLANE0 + c] + ((long)by * 8) * (bw * 8) + bx * 8;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        int o[8];
        idct_1d(ws[8 * r], ws[8 * r + 1], ws[8 * r + 2], ws[8 * r + 3], ws[8 * r + 4], ws[8 * r + 5], ws[8 * r + 6], ws[8 * r + 7], 18, o);
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) { lo |= (uint32_t)range_limit_idct(o[j]) << (8 * j); hi |= (uint32_t)range_limit_idct(o[4 + j]) << (8 * j); }
        uint2 v; v.x = lo; v.y = hi;
        *reinterpret_cast<uint2*>(out + (long)r * (bw * 8)) = v;
    }
}

VR_DEV int clamp255(int x) { return x < 0 ? 0 : x > 255 ? 255 : x; }

// chroma sample at full resolution: jdsample.c fancy upsampling (the plain replication when the component is <= 2 samples wide)
VR_DEV int chroma_at(const uint8_t* p, int stride, int cw, int ch, int x, int y, int hmax, int vmax) {
    if (hmax == 1) return p[(long)y * stride + x];
    const int c = x >> 1;
    if (vmax == 1) {                                       // h2v1
        const int v = p[(long)y * stride + c];
        if (cw <= 2) return v;
        if (!(x & 1)) return c == 0 ? v : (3 * v + p[(long)y * stride + c - 1] + 1) >> 2;
        return c == cw - 1 ? v : (3 * v + p[(long)y * stride + c + 1] + 2) >> 2;
    }
    const int r = y >> 1;                                  // h2v2
    if (cw <= 2) return p[(long)r * stride + c];
    int rf = (y & 1) ? r + 1 : r - 1;                      // the further row; context rows replicate the first / last real row (jdmainct.c)
    rf = rf < 0 ? 0 : rf > ch - 1 ? ch - 1 : rf;
    const uint8_t* n0 = p + (long)r * stride;
    const uint8_t* n1 = p + (long)rf * stride;
    const int cur = 3 * n0[c] + n1[c];
    if (!(x & 1)) return c == 0 ? (cur * 4 + 8) >> 4 : (cur * 3 + 3 * n0[c - 1] + n1[c - 1] + 8) >> 4;
    return c == cw - 1 ? (cur * 4 + 7) >> 4 : (cur * 3 + 3 * n0[c + 1] + n1[c + 1] + 7) >> 4;
}

__global__ __launch_bounds__(256) void jpeg_color_kernel(const long* __restrict__ desc, const uint8_t* __restrict__ planes, uint8_t* __restrict__ rgb) {
    const long* D = desc + (long)blockIdx.y * JD_FIELDS;
    const int W = (int)D[JD_W], H = (int)D[JD_H];
    const long pix = (long)blockIdx.x * 256 + threadIdx.x;
    if (pix >= (long)W * H) return;
    const int y = (int)(pix / W), x = (int)(pix % W);
    const int Y = planes[D[JD_PLANE0] + (long)y * (D[JD_BW0] * 8) + x];
    uint8_t* o = rgb + D[JD_RGB] + pix * 3;
    if (D[JD_NCOMP] == 1) { o[0] = o[1] = o[2] = (uint8_t)Y; return; }
    const int hmax = (int)D[JD_HMAX], vmax = (int)D[JD_VMAX];
    const int cb = chroma_at(planes + D[JD_PLANE1], (int)D[JD_BW1] * 8, (int)D[JD_CW1], (int)D[JD_CH1], x, y, hmax, vmax) - 128;
    const int cr = chroma_at(planes + D[JD_PLANE2], (int)D[JD_BW2] * 8, (int)D[JD_CW2], (int)D[JD_CH2], x, y, hmax, vmax) - 128;
    // jdcolor.c build_ycc_rgb_table / ycc_rgb_convert: SCALEBITS 16, FIX(x) = (int)(x * 65536 + 0.5), arithmetic right shifts
    const int r = Y + ((91881 * cr + 32768) >> 16);
    const int g = Y + ((-22554 * cb + 32768 - 46802 * cr) >> 16);
    const int b = Y + ((116130 * cb + 32768) >> 16);
    o[0] = (uint8_t)clamp255(r); o[1] = (uint8_t)clamp255(g); o[2] = (uint8_t)clamp255(b);
}

}  // namespace

extern "C" int visrep_jpeg_reconstruct(const void* coef, const void* qtab, const void* desc, int n_images, long max_blocks, long max_pixels,
                                       void* planes, void* rgb, void* stream) {
    if (n_images <= 0) return 0;
    if (!coef || !qtab || !desc || !planes || !rgb) return visrep_set_error(VISREP_ERR_ARG, "jpeg_reconstruct: null pointer");
    if (max_blocks <= 0 || max_pixels <= 0 || n_images > 65535) return visrep_set_error(VISREP_ERR_SHAPE, "jpeg_reconstruct: bad batch geometry");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(jpeg_idct_kernel, dim3((unsigned)((max_blocks + 63) / 64), n_images), dim3(64), 0, st, (const int16_t*)coef, (const uint16_t*)qtab,
                       (const long*)desc, (uint8_t*)planes);
    hipLaunchKernelGGL(jpeg_color_kernel, dim3((unsigned)((max_pixels + 255) / 256), n_images), dim3(256), 0, st, (const long*)desc, (const uint8_t*)planes,
                       (uint8_t*)rgb);
    return hipGetLastError() == hipSuccess ? 0 : visrep_set_error(VISREP_ERR_LAUNCH, "jpeg_reconstruct: launch failed");
}
