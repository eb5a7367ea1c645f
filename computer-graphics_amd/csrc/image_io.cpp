// image_io.cpp -- SDL_SaveImage (CW/raytracer/Source/SDLauxiliary.h:25-54) without SDL.
//
// The reference wraps screen->buffer (ARGB8888 words, PutPixelSDL's (128 << 24) | r << 16 | g << 8 | b,
// SDLauxiliary.h:149-161) in an SDL surface with an alpha mask and calls SDL_SaveBMP. For such a
// surface SDL2 writes a 32-bit BI_BITFIELDS bitmap with a BITMAPV4HEADER: 14-byte file header,
// 108-byte info header (masks R 00FF0000, G 0000FF00, B 000000FF, A FF000000, colour space "Win "),
// rows bottom-up, pixel words unchanged. The reference's own screenshot.bmp (500 x 500) has exactly
// this layout; its first 122 bytes are the test fixture for the header.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/mipt.h"

extern "C" int mipt_internal_fail(int code, const char* msg);

namespace {
void put16(std::vector<uint8_t>& b, uint16_t v) { b.push_back((uint8_t)v); b.push_back((uint8_t)(v >> 8)); }
void put32(std::vector<uint8_t>& b, uint32_t v) { put16(b, (uint16_t)v); put16(b, (uint16_t)(v >> 16)); }
}  // namespace

static int save_bmp_impl(const char* path, const uint32_t* screen_buffer, int width, int height) {
  if (!path || !screen_buffer) return mipt_internal_fail(MIPT_E_INVALID, "mipt_save_bmp: null argument");
  if (width <= 0 || height <= 0 || width > 32768 || height > 32768)
    return mipt_internal_fail(MIPT_E_INVALID, "mipt_save_bmp: image size out of range");
  const uint32_t image_bytes = (uint32_t)width * (uint32_t)height * 4u;
  std::vector<uint8_t> h;
  h.reserve(122);
  h.push_back('B'); h.push_back('M');
  put32(h, 122u + image_bytes);  // file size
  put16(h, 0); put16(h, 0);
  put32(h, 122);                 // offset of the pixel data
  put32(h, 108);                 // BITMAPV4HEADER
  put32(h, (uint32_t)width); put32(h, (uint32_t)height);  // positive height: bottom-up
  put16(h, 1); put16(h, 32);
  put32(h, 3);                   // BI_BITFIELDS
  put32(h, image_bytes);
  put32(h, 0); put32(h, 0);      // pixels per metre
  put32(h, 0); put32(h, 0);      // palette
  put32(h, 0x00FF0000u); put32(h, 0x0000FF00u); put32(h, 0x000000FFu); put32(h, 0xFF000000u);
  put32(h, 0x57696E20u);         // LCS_WINDOWS_COLOR_SPACE, "Win "
  for (int k = 0; k < 12; ++k) put32(h, 0);  // endpoints (9) + gamma (3)
  FILE* f = fopen(path, "wb");
  if (!f) return mipt_internal_fail(MIPT_E_INVALID, "mipt_save_bmp: cannot open the file for writing");
  bool ok = fwrite(h.data(), 1, h.size(), f) == h.size();
  for (int y = height - 1; ok && y >= 0; --y)
    ok = fwrite(screen_buffer + (size_t)y * (size_t)width, 4, (size_t)width, f) == (size_t)width;
  ok = (fclose(f) == 0) && ok;
  return ok ? MIPT_OK : mipt_internal_fail(MIPT_E_INVALID, "mipt_save_bmp: write failed");
}

// nothing may throw across the C ABI
extern "C" int mipt_save_bmp(const char* path, const uint32_t* screen_buffer, int width, int height) {
  try {
    return save_bmp_impl(path, screen_buffer, width, height);
  } catch (const std::bad_alloc&) {
    return mipt_internal_fail(MIPT_E_NOMEM, "mipt_save_bmp: host allocation failed");
  } catch (...) {
    return mipt_internal_fail(MIPT_E_INVALID, "mipt_save_bmp: unexpected exception");
  }
}
