// vpt_jpeg.cpp — baseline JPEG writer that reproduces, byte for byte, what the reference's
// save_image(".jpg") emits (yocto_sceneio.cpp:565-571 -> stbi_write_jpg_to_func(..., comp 4,
// quality 75), libs/yocto/ext/stb_image_write.h:1250-1611).  It exists because the ONLY
// external golden images (check/*.jpg) are q75 JPEGs of very noisy renders: the codec moves
// pixels by ~0.03 RMS, so parity against them is only meaningful through the same encoder
// (SURVEY.md fact 11).  Structure is ours (table-driven Huffman built from the JPEG Annex K
// BITS/HUFFVAL lists, block gather -> FDCT -> quantise -> entropy code); the float32 operation
// order of the colour transform, the AAN FDCT and the quantiser follows the reference exactly,
// because a different rounding changes coefficients.  Compile with -ffp-contract=off.
//
// Fixed by quality 75: quality<=90 => 2x2 chroma subsampling (h2v2, 16x16 MCUs, box filter),
// quantiser scale 200-2*75 = 50 (stb :1477-1488).
#include <array>
#include <cstring>

#include "vpt_host.h"

namespace vpt {
namespace {

const uint8_t zigzag[64] = {0, 1, 5, 6, 14, 15, 27, 28, 2, 4, 7, 13, 16, 26, 29, 42, 3, 8, 12, 17,
    25, 30, 41, 43, 9, 11, 18, 24, 31, 40, 44, 53, 10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38,
    46, 51, 55, 60, 21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63};

// JPEG Annex K.1 quantisation tables (natural order)
const int quant_luma[64]   = {16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13,
      16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62, 18, 22, 37, 56, 68, 109, 103, 77, 24, 35,
      55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
const int quant_chroma[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26,
    56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
    99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};

// JPEG Annex K.3 Huffman specifications: BITS (codes per length 1..16) and HUFFVAL
const uint8_t dc_luma_bits[16]   = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
const uint8_t dc_chroma_bits[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
const uint8_t dc_vals[12]        = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
const uint8_t ac_luma_bits[16]   = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
const uint8_t ac_luma_vals[162]  = {0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41,
     0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1,
     0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25,
     0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47,
     0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68,
     0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89,
     0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8,
     0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7,
     0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5,
     0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
const uint8_t ac_chroma_bits[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
const uint8_t ac_chroma_vals[162] = {0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12,
    0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09,
    0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18,
    0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46,
    0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67,
    0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87,
    0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6,
    0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5,
    0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4,
    0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

struct huff_code { uint16_t code = 0, length = 0; };
using huff_table = std::array<huff_code, 256>;

// canonical Huffman code assignment, JPEG Annex C
huff_table make_huffman(const uint8_t bits[16], const uint8_t* vals) {
  auto table = huff_table{};
  auto code = 0u, k = 0u;
  for (auto len = 1u; len <= 16; len++) {
    for (auto i = 0u; i < bits[len - 1]; i++) table[vals[k++]] = {(uint16_t)code++, (uint16_t)len};
    code <<= 1;
  }
  return table;
}

struct bit_writer {
  vector<uint8_t>& out;
  uint32_t         acc = 0;  // bits are left-aligned in a 24-bit window, as in the reference
  int              n   = 0;
  void put(uint16_t code, uint16_t length) {
    n += length;
    acc |= (uint32_t)code << (24 - n);
    while (n >= 8) {
      auto byte = (uint8_t)((acc >> 16) & 255);
      out.push_back(byte);
      if (byte == 255) out.push_back(0);  // byte stuffing
      acc <<= 8;
      n -= 8;
    }
  }
  void put(huff_code c) { put(c.code, c.length); }
};

// magnitude category + extra bits of a coefficient (JPEG F.1.2.1)
huff_code magnitude_bits(int value) {
  auto mag = value < 0 ? -value : value;
  auto low = value < 0 ? value - 1 : value;
  auto len = 1;
  while (mag >>= 1) len++;
  return {(uint16_t)(low & ((1 << len) - 1)), (uint16_t)len};
}

// 8-point AAN forward DCT on v[0], v[s], ..., v[7s] (float32, reference operation order)
void fdct8(float* v, int s) {
  auto d0 = v[0], d1 = v[s], d2 = v[2 * s], d3 = v[3 * s], d4 = v[4 * s], d5 = v[5 * s], d6 = v[6 * s], d7 = v[7 * s];
  auto a0 = d0 + d7, a7 = d0 - d7, a1 = d1 + d6, a6 = d1 - d6;
  auto a2 = d2 + d5, a5 = d2 - d5, a3 = d3 + d4, a4 = d3 - d4;
  // even part
  auto e0 = a0 + a3, e3 = a0 - a3, e1 = a1 + a2, e2 = a1 - a2;
  v[0]     = e0 + e1;
  v[4 * s] = e0 - e1;
  auto r1  = (e2 + e3) * 0.707106781f;
  v[2 * s] = e3 + r1;
  v[6 * s] = e3 - r1;
  // odd part
  auto o0 = a4 + a5, o1 = a5 + a6, o2 = a6 + a7;
  auto r5  = (o0 - o2) * 0.382683433f;
  auto r2  = o0 * 0.541196100f + r5;
  auto r4  = o2 * 1.306562965f + r5;
  auto r3  = o1 * 0.707106781f;
  auto s11 = a7 + r3, s13 = a7 - r3;
  v[5 * s] = s13 + r2;
  v[3 * s] = s13 - r2;
  v[s]     = s11 + r4;
  v[7 * s] = s11 - r4;
}

// FDCT + quantise + entropy-code one 8x8 data unit; returns its DC for prediction
int encode_unit(bit_writer& bw, float* unit, int stride, const float* scale, int dc_pred,
    const huff_table& dc_table, const huff_table& ac_table) {
  for (auto r = 0; r < 8; r++) fdct8(unit + r * stride, 1);
  for (auto c = 0; c < 8; c++) fdct8(unit + c, stride);
  int coef[64];
  for (auto y = 0, j = 0; y < 8; y++)
    for (auto x = 0; x < 8; x++, j++) {
      auto v          = unit[y * stride + x] * scale[j];
      coef[zigzag[j]] = (int)(v < 0 ? v - 0.5f : v + 0.5f);  // round half away, by truncation
    }
  auto diff = coef[0] - dc_pred;
  if (diff == 0) {
    bw.put(dc_table[0]);
  } else {
    auto m = magnitude_bits(diff);
    bw.put(dc_table[m.length]);
    bw.put(m);
  }
  auto last = 63;
  while (last > 0 && coef[last] == 0) last--;
  if (last == 0) {
    bw.put(ac_table[0x00]);  // EOB
    return coef[0];
  }
  for (auto i = 1; i <= last; i++) {
    auto run = 0;
    while (coef[i] == 0 && i <= last) i++, run++;
    for (; run >= 16; run -= 16) bw.put(ac_table[0xF0]);  // ZRL
    auto m = magnitude_bits(coef[i]);
    bw.put(ac_table[(run << 4) + m.length]);
    bw.put(m);
  }
  if (last != 63) bw.put(ac_table[0x00]);
  return coef[0];
}

}  // namespace

vector<uint8_t> encode_jpeg_q75(int width, int height, const vector<vec4b>& rgba) {
  auto out = vector<uint8_t>{};
  if (width <= 0 || height <= 0 || rgba.size() < (size_t)width * height) return out;
  const int quality_scale = 200 - 75 * 2;  // 50

  // quantisation tables in zig-zag order + the AAN-descaled reciprocal multipliers
  uint8_t qt_luma[64], qt_chroma[64];
  for (auto i = 0; i < 64; i++) {
    auto y = (quant_luma[i] * quality_scale + 50) / 100, c = (quant_chroma[i] * quality_scale + 50) / 100;
    qt_luma[zigzag[i]]   = (uint8_t)(y < 1 ? 1 : y > 255 ? 255 : y);
    qt_chroma[zigzag[i]] = (uint8_t)(c < 1 ? 1 : c > 255 ? 255 : c);
  }
  static const float aan[8] = {1.0f * 2.828427125f, 1.387039845f * 2.828427125f,
      1.306562965f * 2.828427125f, 1.175875602f * 2.828427125f, 1.0f * 2.828427125f,
      0.785694958f * 2.828427125f, 0.541196100f * 2.828427125f, 0.275899379f * 2.828427125f};
  float scale_luma[64], scale_chroma[64];
  for (auto row = 0, k = 0; row < 8; row++)
    for (auto col = 0; col < 8; col++, k++) {
      scale_luma[k]   = 1 / (qt_luma[zigzag[k]] * aan[row] * aan[col]);
      scale_chroma[k] = 1 / (qt_chroma[zigzag[k]] * aan[row] * aan[col]);
    }

  auto put  = [&](std::initializer_list<int> bytes) { for (auto b : bytes) out.push_back((uint8_t)b); };
  auto putn = [&](const uint8_t* p, size_t n) { out.insert(out.end(), p, p + n); };
  // SOI, JFIF APP0, DQT (two tables in one segment)
  put({0xFF, 0xD8, 0xFF, 0xE0, 0, 0x10, 'J', 'F', 'I', 'F', 0, 1, 1, 0, 0, 1, 0, 1, 0, 0, 0xFF, 0xDB, 0, 0x84, 0});
  putn(qt_luma, 64);
  put({1});
  putn(qt_chroma, 64);
  // SOF0: 8-bit, 3 components, Y 2x2, Cb/Cr 1x1 ; then one DHT segment with the four tables
  put({0xFF, 0xC0, 0, 0x11, 8, height >> 8, height & 255, width >> 8, width & 255, 3, 1, 0x22, 0, 2, 0x11, 1, 3, 0x11, 1,
      0xFF, 0xC4, 0x01, 0xA2, 0});
  putn(dc_luma_bits, 16), putn(dc_vals, 12);
  put({0x10});
  putn(ac_luma_bits, 16), putn(ac_luma_vals, 162);
  put({1});
  putn(dc_chroma_bits, 16), putn(dc_vals, 12);
  put({0x11});
  putn(ac_chroma_bits, 16), putn(ac_chroma_vals, 162);
  // SOS
  put({0xFF, 0xDA, 0, 0xC, 3, 1, 0, 2, 0x11, 3, 0x11, 0, 0x3F, 0});

  auto dc_l = make_huffman(dc_luma_bits, dc_vals), ac_l = make_huffman(ac_luma_bits, ac_luma_vals);
  auto dc_c = make_huffman(dc_chroma_bits, dc_vals), ac_c = make_huffman(ac_chroma_bits, ac_chroma_vals);
  auto bw   = bit_writer{out};
  auto pred_y = 0, pred_u = 0, pred_v = 0;
  for (auto y0 = 0; y0 < height; y0 += 16)
    for (auto x0 = 0; x0 < width; x0 += 16) {
      float Y[256], U[256], V[256];
      for (auto row = y0, pos = 0; row < y0 + 16; row++) {
        auto rr = row < height ? row : height - 1;  // edge replication
        for (auto col = x0; col < x0 + 16; col++, pos++) {
          auto& px = rgba[(size_t)rr * width + (col < width ? col : width - 1)];
          float r = px.x, g = px.y, b = px.z;
          Y[pos] = +0.29900f * r + 0.58700f * g + 0.11400f * b - 128;
          U[pos] = -0.16874f * r - 0.33126f * g + 0.50000f * b;
          V[pos] = +0.50000f * r - 0.41869f * g - 0.08131f * b;
        }
      }
      pred_y = encode_unit(bw, Y + 0, 16, scale_luma, pred_y, dc_l, ac_l);
      pred_y = encode_unit(bw, Y + 8, 16, scale_luma, pred_y, dc_l, ac_l);
      pred_y = encode_unit(bw, Y + 128, 16, scale_luma, pred_y, dc_l, ac_l);
      pred_y = encode_unit(bw, Y + 136, 16, scale_luma, pred_y, dc_l, ac_l);
      float sub_u[64], sub_v[64];
      for (auto yy = 0, pos = 0; yy < 8; yy++)
        for (auto xx = 0; xx < 8; xx++, pos++) {
          auto j     = yy * 32 + xx * 2;
          sub_u[pos] = (U[j] + U[j + 1] + U[j + 16] + U[j + 17]) * 0.25f;
          sub_v[pos] = (V[j] + V[j + 1] + V[j + 16] + V[j + 17]) * 0.25f;
        }
      pred_u = encode_unit(bw, sub_u, 8, scale_chroma, pred_u, dc_c, ac_c);
      pred_v = encode_unit(bw, sub_v, 8, scale_chroma, pred_v, dc_c, ac_c);
    }
  bw.put(0x7F, 7);  // pad to a byte boundary with ones
  put({0xFF, 0xD9});
  return out;
}

}  // namespace vpt
