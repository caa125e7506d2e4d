// JPEG entropy decoding on the host (the part of `Image.open(path).convert('RGB')`,
// ref/modules/loader.py:15, that is inherently serial): marker parsing and Huffman decoding of baseline / extended-sequential /
// progressive 8-bit JPEG (SOF0 / SOF1 / SOF2) into quantised DCT coefficient blocks.  Everything after it -- dequantisation, the inverse DCT,
// chroma upsampling and the colour transform -- runs on the GPU (csrc/jpeg.hip) and reproduces libjpeg-turbo's decoder (the one
// Pillow links) bit for bit.  Written from the JPEG standard (ITU-T T.81): Annex B (markers), C (table construction), F.2.2
// (sequential Huffman decoding), G.1.2 (progressive: spectral selection and successive approximation), figure A.6 (zig-zag order).
#include <atomic>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "klab_mm.h"

namespace {

const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huff {
  bool defined = false;
  uint8_t bits[17] = {0};
  uint8_t vals[256] = {0};
  // decoding tables (T.81 F.2.2.3): 9-bit lookahead, then the canonical maxcode / valptr walk
  int16_t look_sym[512];
  uint8_t look_len[512];
  int32_t maxcode[18];
  int32_t valoff[17];
  bool build() {
    int n = 0;
    for (int l = 1; l <= 16; ++l) n += bits[l];
    if (n > 256) return false;
    uint32_t code = 0;
    int k = 0;
    memset(look_len, 0, sizeof(look_len));
    for (int l = 1; l <= 16; ++l) {
      valoff[l] = k - (int)code;
      for (int i = 0; i < bits[l]; ++i, ++k, ++code) {
        if (code >= (1u << l)) return false;
        if (l <= 9) {
          const int lo = (int)(code << (9 - l)), cnt = 1 << (9 - l);
          for (int j = 0; j < cnt; ++j) { look_sym[lo + j] = vals[k]; look_len[lo + j] = (uint8_t)l; }
        }
      }
      maxcode[l] = bits[l] ? (int32_t)code - 1 : -1;
      code <<= 1;
    }
    maxcode[17] = 0x7fffffff;
    defined = true;
    return true;
  }
};

struct Comp { int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0; int bw = 0, bh = 0; long long block0 = 0; };

struct Parsed {
  int width = 0, height = 0, ncomp = 0, precision = 8, hmax = 1, vmax = 1, mcus_x = 0, mcus_y = 0;
  bool progressive = false, have_sof = false, jfif = false, adobe = false;
  int n_scans = 0;
  int adobe_transform = -1;
  int restart_interval = 0;
  Comp comp[4];
  uint16_t qt[4][64];
  bool qt_defined[4] = {false, false, false, false};
  Huff dc[4], ac[4];
};

struct Reader {
  const uint8_t* p;
  const uint8_t* end;
  uint64_t buf = 0;
  int nbits = 0;
  bool hit_marker = false;
  void reset() { buf = 0; nbits = 0; hit_marker = false; }
  inline void fill() {
    // fast path: four data bytes at once when none of them is 0xFF (no stuffing, no marker); afterwards nbits >= 32
    if (!hit_marker && nbits <= 32 && p + 4 <= end) {
      const uint32_t w = ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
      const uint32_t inv = ~w;
      if (((inv - 0x01010101u) & w & 0x80808080u) == 0) {  // no byte of ~w is zero
        buf |= (uint64_t)w << (32 - nbits);
        nbits += 32;
        p += 4;
        return;
      }
    }
    while (nbits <= 56) {
      uint32_t b = 0;
      if (!hit_marker && p < end) {
        b = *p;
        if (b == 0xFF) {
          if (p + 1 < end && p[1] == 0x00) p += 2;            // stuffed zero
          else if (p + 1 < end && p[1] == 0xFF) { ++p; continue; }  // fill byte
          else { hit_marker = true; b = 0; }                  // a marker: feed zeros (as libjpeg does on truncated data)
        } else ++p;
      } else { hit_marker = true; }
      buf |= (uint64_t)b << (56 - nbits);
      nbits += 8;
    }
  }
  inline uint32_t peek(int n) { return (uint32_t)(buf >> (64 - n)); }
  inline void drop(int n) { buf <<= n; nbits -= n; }
  inline int receive_extend(int s) {
    if (s == 0) return 0;
    if (nbits < s) fill();
    const int v = (int)peek(s);
    drop(s);
    return v - ((((v >> (s - 1)) & 1) - 1) & ((1 << s) - 1));  // v < 2^(s-1): v - (2^s - 1)
  }
  inline int decode(const Huff& h) {
    if (nbits < 32) fill();  // a code (<= 16 bits) and the value bits behind it (<= 15) without another refill
    const uint32_t look = peek(9);
    int l = h.look_len[look];
    if (l) { drop(l); return h.look_sym[look]; }
    l = 10;
    int32_t code = (int32_t)peek(10);
    while (l <= 16 && code > h.maxcode[l]) { ++l; code = (int32_t)peek(l); }
    if (l > 16) return -1;
    drop(l);
    return h.vals[(code + h.valoff[l]) & 255];
  }
};

inline int rd16(const uint8_t* p) { return (p[0] << 8) | p[1]; }

int geometry(Parsed& P) {
  if (P.width <= 0 || P.height <= 0 || P.ncomp < 1) return KLAB_ERR_BADARG;
  P.hmax = P.vmax = 1;
  for (int c = 0; c < P.ncomp; ++c) { P.hmax = P.comp[c].h > P.hmax ? P.comp[c].h : P.hmax; P.vmax = P.comp[c].v > P.vmax ? P.comp[c].v : P.vmax; }
  P.mcus_x = (P.width + 8 * P.hmax - 1) / (8 * P.hmax);
  P.mcus_y = (P.height + 8 * P.vmax - 1) / (8 * P.vmax);
  long long b = 0;
  for (int c = 0; c < P.ncomp; ++c) {
    P.comp[c].bw = P.mcus_x * P.comp[c].h;
    P.comp[c].bh = P.mcus_y * P.comp[c].v;
    P.comp[c].block0 = b;
    b += (long long)P.comp[c].bw * P.comp[c].bh;
  }
  return KLAB_OK;
}

// Walk the markers.  coefs == nullptr: stop at the first SOS (header only).
int parse(const uint8_t* data, size_t n, Parsed& P, int16_t* coefs) {
  if (!data || n < 4 || data[0] != 0xFF || data[1] != 0xD8) return KLAB_ERR_BADARG;
  const uint8_t* p = data + 2;
  const uint8_t* end = data + n;
  bool decoded_any = false;
  while (p + 4 <= end) {
    if (*p != 0xFF) { ++p; continue; }
    while (p < end && *p == 0xFF) ++p;
    if (p >= end) break;
    const int m = *p++;
    if (m == 0xD9) break;                           // EOI
    if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;  // stand-alone markers
    if (p + 2 > end) return KLAB_ERR_BADARG;
    const int len = rd16(p);
    if (len < 2 || p + len > end) return KLAB_ERR_BADARG;
    const uint8_t* s = p + 2;
    const uint8_t* se = p + len;
    if (m == 0xC0 || m == 0xC1 || m == 0xC2) {
      if (len < 8 || P.have_sof) return KLAB_ERR_BADARG;  // (a second frame header would change the geometry the caller sized its buffer for)
      P.progressive = (m == 0xC2);
      P.precision = s[0];
      P.height = rd16(s + 1);
      P.width = rd16(s + 3);
      P.ncomp = s[5];
      if (P.ncomp < 1 || P.ncomp > 4 || len < 8 + 3 * P.ncomp) return KLAB_ERR_BADARG;
      for (int c = 0; c < P.ncomp; ++c) {
        P.comp[c].id = s[6 + 3 * c];
        P.comp[c].h = s[7 + 3 * c] >> 4;
        P.comp[c].v = s[7 + 3 * c] & 15;
        P.comp[c].tq = s[8 + 3 * c] & 3;
        if (P.comp[c].h < 1 || P.comp[c].h > 4 || P.comp[c].v < 1 || P.comp[c].v > 4) return KLAB_ERR_BADARG;
      }
      // untrusted files (RedCaps images come from the web): refuse frames beyond Pillow's own decompression-bomb limit
      // (Image.MAX_IMAGE_PIXELS * 2 = 178,956,970 pixels raises DecompressionBombError in the reference's Image.open) before
      // anybody sizes a buffer from the header
      if (P.width < 1 || P.height < 1 || (long long)P.width * P.height > 178956970LL) return KLAB_ERR_UNSUPPORTED;
      P.have_sof = true;
      const int rc = geometry(P);
      if (rc) return rc;
    } else if (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
      return KLAB_ERR_UNSUPPORTED;  // lossless / hierarchical / arithmetic
    } else if (m == 0xDB) {
      while (s < se) {
        const int pq = s[0] >> 4, tq = s[0] & 15;
        ++s;
        if (tq > 3 || pq > 1 || s + (pq ? 128 : 64) > se) return KLAB_ERR_BADARG;
        for (int i = 0; i < 64; ++i) { P.qt[tq][kZigzag[i]] = (uint16_t)(pq ? rd16(s + 2 * i) : s[i]); }
        s += pq ? 128 : 64;
        P.qt_defined[tq] = true;
      }
    } else if (m == 0xC4) {
      while (s < se) {
        if (s + 17 > se) return KLAB_ERR_BADARG;
        const int tc = s[0] >> 4, th = s[0] & 15;
        if (tc > 1 || th > 3) return KLAB_ERR_BADARG;
        Huff& h = tc ? P.ac[th] : P.dc[th];
        int cnt = 0;
        h.bits[0] = 0;
        for (int l = 1; l <= 16; ++l) { h.bits[l] = s[l]; cnt += s[l]; }
        s += 17;
        if (cnt > 256 || s + cnt > se) return KLAB_ERR_BADARG;
        memcpy(h.vals, s, cnt);
        s += cnt;
        if (!h.build()) return KLAB_ERR_BADARG;
      }
    } else if (m == 0xDD) {
      if (len < 4) return KLAB_ERR_BADARG;
      P.restart_interval = rd16(s);
    } else if (m == 0xE0) {
      if (len >= 7 && !memcmp(s, "JFIF", 5)) P.jfif = true;
    } else if (m == 0xEE) {
      if (len >= 14 && !memcmp(s, "Adobe", 5)) { P.adobe = true; P.adobe_transform = s[11]; }
    } else if (m == 0xDA) {
      if (!P.have_sof) return KLAB_ERR_BADARG;
      if (!coefs) return KLAB_OK;  // header only
      if (P.precision != 8) return KLAB_ERR_UNSUPPORTED;
      if (len < 6) return KLAB_ERR_BADARG;  // (a segment cut right behind its length field: s == se, nothing to read)
      if (++P.n_scans > 1024) return KLAB_ERR_UNSUPPORTED;  // a progressive file needs a few dozen scans; bound the work a crafted one can ask for
      const int ns = s[0];
      if (ns < 1 || ns > P.ncomp || len < 6 + 2 * ns) return KLAB_ERR_BADARG;
      const int Ss = s[1 + 2 * ns], Se = s[2 + 2 * ns], Ah = s[3 + 2 * ns] >> 4, Al = s[3 + 2 * ns] & 15;
      // scan kind: 0 sequential; progressive (T.81 G.1.2): 1 DC first, 2 DC refinement, 3 AC first, 4 AC refinement
      int kind = 0;
      if (P.progressive) {
        if (Ss > Se || Se > 63 || Al > 13 || (Ss == 0 && Se != 0) || (Ss > 0 && ns != 1) || (Ah != 0 && Ah != Al + 1)) return KLAB_ERR_BADARG;
        kind = Ss == 0 ? (Ah == 0 ? 1 : 2) : (Ah == 0 ? 3 : 4);
      } else if (ns != P.ncomp) {
        // sequential files: one interleaved scan over whole MCUs (what every baseline encoder in common use writes); a file that
        // codes its components in separate scans is reported as unsupported rather than taken down a path no test file exercises
        return KLAB_ERR_UNSUPPORTED;
      }
      int sc[4];
      for (int i = 0; i < ns; ++i) {
        const int cid = s[1 + 2 * i];
        int ci = -1;
        for (int c = 0; c < P.ncomp; ++c) if (P.comp[c].id == cid) ci = c;
        if (ci < 0) return KLAB_ERR_BADARG;
        P.comp[ci].td = s[2 + 2 * i] >> 4;
        P.comp[ci].ta = s[2 + 2 * i] & 15;
        if (P.comp[ci].td > 3 || P.comp[ci].ta > 3) return KLAB_ERR_BADARG;
        if ((kind == 0 || kind == 1) && !P.dc[P.comp[ci].td].defined) return KLAB_ERR_BADARG;
        if ((kind == 0 || kind >= 3) && !P.ac[P.comp[ci].ta].defined) return KLAB_ERR_BADARG;
        sc[i] = ci;
      }
      // ---- entropy-coded segment -------------------------------------------------------------
      Reader r{se, end};
      int pred[4] = {0, 0, 0, 0};
      int eobrun = 0;
      // an interleaved scan runs over MCUs, a one-component scan over that component's own block grid (T.81 A.2.2)
      int mx = P.mcus_x, my = P.mcus_y;
      if (ns == 1) {
        const Comp& c = P.comp[sc[0]];
        mx = ((P.width * c.h + P.hmax - 1) / P.hmax + 7) / 8;
        my = ((P.height * c.v + P.vmax - 1) / P.vmax + 7) / 8;
      }
      const int p1 = 1 << Al, m1 = -(1 << Al);
      int until_restart = P.restart_interval, next_rst = 0;
      for (int y = 0; y < my; ++y)
        for (int x = 0; x < mx; ++x) {
          if (P.restart_interval && until_restart == 0) {
            // byte-align, find RSTn
            r.reset();
            const uint8_t* q = r.p;
            while (q + 1 < end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) {
              if (q[0] == 0xFF && q[1] != 0x00 && q[1] != 0xFF) break;  // some other marker: give up resynchronising
              ++q;
            }
            if (q + 1 < end && q[0] == 0xFF && q[1] == 0xD0 + next_rst) q += 2;
            next_rst = (next_rst + 1) & 7;
            r.p = q;
            pred[0] = pred[1] = pred[2] = pred[3] = 0;
            eobrun = 0;
            until_restart = P.restart_interval;
          }
          for (int i = 0; i < ns; ++i) {
            const Comp& c = P.comp[sc[i]];
            const int nh = ns == 1 ? 1 : c.h, nv = ns == 1 ? 1 : c.v;
            for (int by = 0; by < nv; ++by)
              for (int bx = 0; bx < nh; ++bx) {
                const int gx = x * nh + bx, gy = y * nv + by;
                int16_t* blk = coefs + (c.block0 + (long long)gy * c.bw + gx) * 64;
                if (kind == 0 || kind == 1) {  // DC coefficient: Huffman-coded difference to the previous block of the component
                  const int sdc = r.decode(P.dc[c.td]);
                  if (sdc < 0 || sdc > 15) return KLAB_ERR_BADARG;
                  pred[sc[i]] += r.receive_extend(sdc);
                  blk[0] = (int16_t)(kind == 1 ? pred[sc[i]] * (1 << Al) : pred[sc[i]]);
                }
                if (kind == 0) {
                  const Huff& ha = P.ac[c.ta];
                  for (int k = 1; k < 64;) {
                    const int rs = r.decode(ha);
                    if (rs < 0) return KLAB_ERR_BADARG;
                    const int run = rs >> 4, sz = rs & 15;
                    if (sz == 0) {
                      if (run != 15) break;
                      k += 16;
                      continue;
                    }
                    k += run;
                    if (k > 63) return KLAB_ERR_BADARG;
                    blk[kZigzag[k]] = (int16_t)r.receive_extend(sz);
                    ++k;
                  }
                } else if (kind == 2) {  // DC refinement: one more bit of every DC coefficient
                  if (r.nbits < 1) r.fill();
                  if (r.peek(1)) blk[0] = (int16_t)(blk[0] | p1);
                  r.drop(1);
                } else if (kind == 3) {  // AC first pass of the band Ss..Se, with end-of-band runs over blocks (G.1.2.2)
                  if (eobrun > 0) { --eobrun; continue; }
                  const Huff& ha = P.ac[c.ta];
                  for (int k = Ss; k <= Se; ++k) {
                    const int rs = r.decode(ha);
                    if (rs < 0) return KLAB_ERR_BADARG;
                    const int run = rs >> 4, sz = rs & 15;
                    if (sz) {
                      k += run;
                      if (k > 63) return KLAB_ERR_BADARG;
                      blk[kZigzag[k]] = (int16_t)(r.receive_extend(sz) * (1 << Al));
                    } else if (run == 15) {
                      k += 15;
                    } else {
                      eobrun = 1 << run;
                      if (run) { if (r.nbits < run) r.fill(); eobrun += (int)r.peek(run); r.drop(run); }
                      --eobrun;
                      break;
                    }
                  }
                } else if (kind == 4) {  // AC refinement (G.1.2.3): correction bits for the non-zero history, newly non-zero coefficients of +-2^Al
                  const Huff& ha = P.ac[c.ta];
                  int k = Ss;
                  auto refine = [&](int16_t& cf) {
                    if (r.nbits < 1) r.fill();
                    const int bit = (int)r.peek(1);
                    r.drop(1);
                    if (bit && (cf & p1) == 0) cf = (int16_t)(cf >= 0 ? cf + p1 : cf + m1);
                  };
                  if (eobrun == 0) {
                    for (; k <= Se; ++k) {
                      const int rs = r.decode(ha);
                      if (rs < 0) return KLAB_ERR_BADARG;
                      int run = rs >> 4, sz = rs & 15, val = 0;
                      if (sz) {
                        if (sz != 1) return KLAB_ERR_BADARG;
                        if (r.nbits < 1) r.fill();
                        val = r.peek(1) ? p1 : m1;
                        r.drop(1);
                      } else if (run != 15) {
                        eobrun = 1 << run;
                        if (run) { if (r.nbits < run) r.fill(); eobrun += (int)r.peek(run); r.drop(run); }
                        break;  // the rest of the band is handled by the end-of-band branch below
                      }
                      // skip `run` zero-history coefficients, refining the non-zero ones passed on the way
                      for (; k <= Se; ++k) {
                        int16_t& cf = blk[kZigzag[k]];
                        if (cf != 0) refine(cf);
                        else if (--run < 0) break;
                      }
                      if (val && k <= Se) blk[kZigzag[k]] = (int16_t)val;
                    }
                  }
                  if (eobrun > 0) {
                    for (; k <= Se; ++k) {
                      int16_t& cf = blk[kZigzag[k]];
                      if (cf != 0) refine(cf);
                    }
                    --eobrun;
                  }
                }
              }
          }
          if (P.restart_interval) --until_restart;
        }
      decoded_any = true;
      // continue behind the entropy-coded data: the reader stopped in front of the next marker (or ran out of data)
      const uint8_t* q = r.p;
      while (q + 1 < end && !(q[0] == 0xFF && q[1] != 0x00 && q[1] != 0xFF && !(q[1] >= 0xD0 && q[1] <= 0xD7))) ++q;
      p = q;
      continue;
    }
    p += len;
  }
  if (coefs && !decoded_any) return KLAB_ERR_BADARG;
  return P.have_sof ? KLAB_OK : KLAB_ERR_BADARG;
}

int colour_of(const Parsed& P) {  // libjpeg's default_decompress_parms: how the three components map to RGB
  if (P.ncomp == 1) return KLAB_JPEG_GRAY;
  if (P.ncomp != 3) return -1;
  if (P.jfif) return KLAB_JPEG_YCC;
  if (P.adobe) return P.adobe_transform == 0 ? KLAB_JPEG_RGB : KLAB_JPEG_YCC;
  if (P.comp[0].id == 'R' && P.comp[1].id == 'G' && P.comp[2].id == 'B') return KLAB_JPEG_RGB;
  return KLAB_JPEG_YCC;
}

int fill_info(const Parsed& P, klab_jpeg_info* o) {
  memset(o, 0, sizeof(*o));
  o->width = P.width; o->height = P.height; o->ncomp = P.ncomp; o->progressive = P.progressive ? 1 : 0; o->precision = P.precision;
  o->hmax = P.hmax; o->vmax = P.vmax; o->mcus_x = P.mcus_x; o->mcus_y = P.mcus_y;
  long long b = 0;
  for (int c = 0; c < P.ncomp && c < 3; ++c) {
    o->hs[c] = P.comp[c].h; o->vs[c] = P.comp[c].v; o->bw[c] = P.comp[c].bw; o->bh[c] = P.comp[c].bh; o->tq[c] = P.comp[c].tq;
    b += (long long)P.comp[c].bw * P.comp[c].bh;
  }
  o->coef_blocks = b;
  o->colour = colour_of(P);
  return KLAB_OK;
}

bool supported(const Parsed& P) {
  if (P.precision != 8 || (P.ncomp != 1 && P.ncomp != 3)) return false;
  if (P.ncomp == 3) {
    // luma at full resolution, both chroma planes with the same factors, 1x1 / 2x1 / 2x2 sub-sampling
    if (P.comp[0].h != P.hmax || P.comp[0].v != P.vmax) return false;
    if (P.comp[1].h != P.comp[2].h || P.comp[1].v != P.comp[2].v || P.comp[1].h != 1 || P.comp[1].v != 1) return false;
    if (!((P.hmax == 1 && P.vmax == 1) || (P.hmax == 2 && P.vmax == 1) || (P.hmax == 2 && P.vmax == 2))) return false;
  }
  return true;
}

}  // namespace

extern "C" int klab_jpeg_read_info(const unsigned char* data, size_t n, klab_jpeg_info* info) {
  if (!info) return KLAB_ERR_BADARG;
  Parsed P;
  const int rc = parse(data, n, P, nullptr);
  if (rc) return rc;
  fill_info(P, info);
  info->supported = supported(P) ? 1 : 0;
  return KLAB_OK;
}

extern "C" int klab_jpeg_entropy_decode(const unsigned char* data, size_t n, short* coefs, unsigned short* qt, klab_jpeg_info* info) {
  if (!coefs || !qt) return KLAB_ERR_BADARG;
  Parsed P;
  int rc = parse(data, n, P, nullptr);
  if (rc) return rc;
  if (!supported(P)) return KLAB_ERR_UNSUPPORTED;
  long long blocks = 0;
  for (int c = 0; c < P.ncomp; ++c) blocks += (long long)P.comp[c].bw * P.comp[c].bh;
  memset(coefs, 0, (size_t)blocks * 64 * sizeof(short));
  Parsed Q;  // second pass decodes (tables may be redefined between scans, so everything is re-walked in order)
  rc = parse(data, n, Q, coefs);
  if (rc) return rc;
  for (int c = 0; c < 3; ++c) {
    const int t = c < Q.ncomp ? Q.comp[c].tq : 0;
    if (c < Q.ncomp && !Q.qt_defined[t]) return KLAB_ERR_BADARG;
    for (int i = 0; i < 64; ++i) qt[c * 64 + i] = c < Q.ncomp ? Q.qt[t][i] : 0;
  }
  if (info) { fill_info(Q, info); info->supported = 1; }
  return KLAB_OK;
}

extern "C" int klab_jpeg_entropy_decode_batch(const unsigned char* const* data, const size_t* sizes, int n, short* const* coefs,
                                              unsigned short* qt, klab_jpeg_info* infos, int* rcs, int n_threads) {
  if (!data || !sizes || !coefs || !qt || n < 0) return KLAB_ERR_BADARG;
  if (n_threads < 1) n_threads = 1;
  if (n_threads > n) n_threads = n > 0 ? n : 1;
  std::atomic<int> next{0}, worst{0};
  auto work = [&]() {
    for (;;) {
      const int i = next.fetch_add(1);
      if (i >= n) break;
      const int rc = klab_jpeg_entropy_decode(data[i], sizes[i], coefs[i], qt + (size_t)i * 192, infos ? infos + i : nullptr);
      if (rcs) rcs[i] = rc;
      if (rc) worst.store(rc);
    }
  };
  if (n_threads == 1) work();
  else {
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; ++t) th.emplace_back(work);
    for (auto& t : th) t.join();
  }
  return worst.load();
}
