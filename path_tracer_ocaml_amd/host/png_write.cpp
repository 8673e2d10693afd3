// png_write.cpp -- minimal PNG encoder (8-bit RGB, zlib "stored" blocks) for Render_command's save step
// (render_command/src/render_command.ml:66-70,107).  f64 -> u8 by truncation of v*255, clamped: this is
// what reproduces the reference's shirley-spheres.png byte for byte (tests/test_oracle_golden.py).
#include <cstdint>
#include <cstdio>
#include <vector>

#include "host.h"

namespace {
uint32_t crc_table[256];
bool crc_ready = false;
void crc_init() {
  for (uint32_t n = 0; n < 256; ++n) {
    uint32_t c = n;
    for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xedb88320u ^ (c >> 1) : c >> 1;
    crc_table[n] = c;
  }
  crc_ready = true;
}
uint32_t crc32(const uint8_t* p, size_t n, uint32_t c = 0xffffffffu) {
  if (!crc_ready) crc_init();
  for (size_t i = 0; i < n; ++i) c = crc_table[(c ^ p[i]) & 0xff] ^ (c >> 8);
  return c;
}
void put32(std::vector<uint8_t>& v, uint32_t x) {
  for (int s = 24; s >= 0; s -= 8) v.push_back((uint8_t)(x >> s));
}
void chunk(std::vector<uint8_t>& out, const char type[4], const std::vector<uint8_t>& data) {
  put32(out, (uint32_t)data.size());
  std::vector<uint8_t> body(type, type + 4);
  body.insert(body.end(), data.begin(), data.end());
  out.insert(out.end(), body.begin(), body.end());
  put32(out, crc32(body.data(), body.size()) ^ 0xffffffffu);
}
}  // namespace

extern "C" int32_t pth_write_png(const char* path, int32_t width, int32_t height, const double* rgb) {
  if (!path || !rgb || width <= 0 || height <= 0) return -1;
  std::vector<uint8_t> raw;
  raw.reserve((size_t)height * ((size_t)width * 3 + 1));
  for (int y = 0; y < height; ++y) {
    raw.push_back(0);  // filter: none
    for (int x = 0; x < width * 3; ++x) {
      double v = rgb[(size_t)y * width * 3 + x] * 255.0;
      if (!(v > 0.0)) v = 0.0;
      if (v > 255.0) v = 255.0;
      raw.push_back((uint8_t)(int)v);
    }
  }
  std::vector<uint8_t> z = {0x78, 0x01};
  uint32_t a = 1, b = 0;  // adler32
  for (uint8_t c : raw) {
    a = (a + c) % 65521u;
    b = (b + a) % 65521u;
  }
  for (size_t off = 0; off < raw.size(); off += 65535) {
    const size_t n = raw.size() - off < 65535 ? raw.size() - off : 65535;
    z.push_back(off + n == raw.size() ? 1 : 0);
    z.push_back((uint8_t)(n & 0xff));
    z.push_back((uint8_t)(n >> 8));
    z.push_back((uint8_t)(~n & 0xff));
    z.push_back((uint8_t)((~n >> 8) & 0xff));
    z.insert(z.end(), raw.begin() + (long)off, raw.begin() + (long)(off + n));
  }
  put32(z, (b << 16) | a);
  std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  std::vector<uint8_t> ihdr;
  put32(ihdr, (uint32_t)width);
  put32(ihdr, (uint32_t)height);
  ihdr.insert(ihdr.end(), {8, 2, 0, 0, 0});
  chunk(out, "IHDR", ihdr);
  chunk(out, "IDAT", z);
  chunk(out, "IEND", {});
  FILE* f = std::fopen(path, "wb");
  if (!f) return -2;
  const size_t w = std::fwrite(out.data(), 1, out.size(), f);
  std::fclose(f);
  return w == out.size() ? 0 : -3;
}
