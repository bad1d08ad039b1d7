// Host side of the S-NeRF++ frame writer: a multi-threaded PNG encoder for the quantised frame buffers
// (8-bit RGB / label images, 16-bit depth) that s-nerfpp/zipnerf/random_render_waymo_seq.py:214-227 writes through PIL
// (Image.fromarray(...).save(path)) and that the foreground stages read back (stage1_code/utils_render.py:51-73).
// PNG is lossless: the contract is that a decoder returns exactly the pixels handed in.
//
// Layout: rows are split into one strip per thread; each strip is filtered (per-row choice among None / Sub / Up / Paeth by the
// minimum-sum-of-absolute-differences heuristic) and compressed as a raw deflate stream that ends on a byte boundary
// (Z_SYNC_FLUSH; Z_FINISH for the last strip), so that the strips concatenate into one valid zlib stream; the Adler-32 of the whole
// is combined from the per-strip checksums.  One IDAT chunk.
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../../../include/snerf_io.h"

namespace {

struct Strip {
  std::vector<unsigned char> out;
  uLong adler = 1;
  uLong raw_len = 0;
  int status = Z_OK;
};

inline int paeth(int a, int b, int c) {
  const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

inline long sad(const unsigned char* v, int n) {
  long s = 0;
  for (int i = 0; i < n; ++i) s += std::abs((int)(signed char)v[i]);
  return s;
}

// filters `row` (with `prev` = previous raw row or null) into dst[1..], dst[0] = filter type.  Each candidate is one tight loop
// (the compiler vectorises None / Sub / Up) into its own scratch row; the cheapest by sum of absolute values wins.
void filter_row(const unsigned char* row, const unsigned char* prev, int n, int bpp, unsigned char* dst, unsigned char* scratch) {
  unsigned char* sub = scratch;
  unsigned char* up = scratch + n;
  unsigned char* pae = scratch + 2 * (size_t)n;
  for (int i = 0; i < bpp && i < n; ++i) sub[i] = row[i];
  for (int i = bpp; i < n; ++i) sub[i] = (unsigned char)(row[i] - row[i - bpp]);
  long best = sad(row, n);
  const unsigned char* pick = row;
  int type = 0;
  long s = sad(sub, n);
  if (s < best) { best = s; pick = sub; type = 1; }
  if (prev != nullptr) {
    for (int i = 0; i < n; ++i) up[i] = (unsigned char)(row[i] - prev[i]);
    s = sad(up, n);
    if (s < best) { best = s; pick = up; type = 2; }
    for (int i = 0; i < bpp && i < n; ++i) pae[i] = (unsigned char)(row[i] - prev[i]);
    for (int i = bpp; i < n; ++i) pae[i] = (unsigned char)(row[i] - paeth(row[i - bpp], prev[i], prev[i - bpp]));
    s = sad(pae, n);
    if (s < best) { best = s; pick = pae; type = 4; }
  }
  dst[0] = (unsigned char)type;
  std::memcpy(dst + 1, pick, (size_t)n);
}

void put32(std::vector<unsigned char>& v, uint32_t x) {
  v.push_back((unsigned char)(x >> 24)); v.push_back((unsigned char)(x >> 16)); v.push_back((unsigned char)(x >> 8)); v.push_back((unsigned char)x);
}

void chunk(std::vector<unsigned char>& png, const char* type, const unsigned char* data, size_t len) {
  put32(png, (uint32_t)len);
  const size_t start = png.size();
  png.insert(png.end(), type, type + 4);
  if (len) png.insert(png.end(), data, data + len);
  put32(png, (uint32_t)crc32(0L, png.data() + start, (uInt)(len + 4)));
}

int encode(const void* pixels, int width, int height, int channels, int bit_depth, int level, int threads, std::vector<unsigned char>& png) {
  if (pixels == nullptr || width <= 0 || height <= 0 || channels < 1 || channels > 4 || (bit_depth != 8 && bit_depth != 16)) return SNERF_IO_ERR_ARG;
  if (level < 0 || level > 9) level = 6;
  const int bps = bit_depth / 8, bpp = channels * bps;
  const long row_bytes = (long)width * bpp;
  if (row_bytes > (1L << 30)) return SNERF_IO_ERR_ARG;
  int T = threads < 1 ? 1 : threads;
  if (T > height) T = height;
  if (T > 64) T = 64;
  // 16-bit samples are big-endian in the file: swap once into a private copy
  std::vector<unsigned char> swapped;
  const unsigned char* src = (const unsigned char*)pixels;
  if (bps == 2) {
    swapped.resize((size_t)row_bytes * height);
    const unsigned char* p = src;
    for (size_t i = 0; i + 1 < swapped.size(); i += 2) { swapped[i] = p[i + 1]; swapped[i + 1] = p[i]; }
    src = swapped.data();
  }
  std::vector<Strip> strips(T);
  auto work = [&](int t) {
    const int r0 = (int)((long)height * t / T), r1 = (int)((long)height * (t + 1) / T);
    Strip& s = strips[t];
    const size_t flen = (size_t)(r1 - r0) * (row_bytes + 1);
    std::vector<unsigned char> filt(flen), scratch((size_t)row_bytes * 3);
    for (int r = r0; r < r1; ++r)
      filter_row(src + (size_t)r * row_bytes, r > 0 ? src + (size_t)(r - 1) * row_bytes : nullptr, (int)row_bytes, bpp,
                 filt.data() + (size_t)(r - r0) * (row_bytes + 1), scratch.data());
    s.adler = adler32(1L, filt.data(), (uInt)flen);
    s.raw_len = (uLong)flen;
    z_stream z;
    std::memset(&z, 0, sizeof(z));
    if (deflateInit2(&z, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) { s.status = Z_MEM_ERROR; return; }
    s.out.resize(deflateBound(&z, (uLong)flen) + 16);
    z.next_in = filt.data(); z.avail_in = (uInt)flen;
    z.next_out = s.out.data(); z.avail_out = (uInt)s.out.size();
    const int rc = deflate(&z, t == T - 1 ? Z_FINISH : Z_SYNC_FLUSH);
    if ((t == T - 1 && rc != Z_STREAM_END) || (t != T - 1 && (rc != Z_OK || z.avail_in != 0))) s.status = Z_BUF_ERROR;
    s.out.resize(s.out.size() - z.avail_out);
    deflateEnd(&z);
  };
  if (T == 1) work(0);
  else {
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back(work, t);
    for (auto& x : th) x.join();
  }
  size_t total = 2 + 4;
  for (auto& s : strips) { if (s.status != Z_OK) return SNERF_IO_ERR_ZLIB; total += s.out.size(); }
  std::vector<unsigned char> idat;
  idat.reserve(total);
  idat.push_back(0x78); idat.push_back(0x9C);               // zlib header: deflate, 32 KiB window, default level, no dictionary
  uLong ad = strips[0].adler;
  idat.insert(idat.end(), strips[0].out.begin(), strips[0].out.end());
  for (int t = 1; t < T; ++t) {
    ad = adler32_combine(ad, strips[t].adler, (z_off_t)strips[t].raw_len);
    idat.insert(idat.end(), strips[t].out.begin(), strips[t].out.end());
  }
  put32(idat, (uint32_t)ad);
  png.clear();
  png.reserve(idat.size() + 64);
  static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  png.insert(png.end(), sig, sig + 8);
  std::vector<unsigned char> ihdr;
  put32(ihdr, (uint32_t)width); put32(ihdr, (uint32_t)height);
  static const unsigned char ctype[5] = {0, 0, 4, 2, 6};
  ihdr.push_back((unsigned char)bit_depth); ihdr.push_back(ctype[channels]); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
  chunk(png, "IHDR", ihdr.data(), ihdr.size());
  chunk(png, "IDAT", idat.data(), idat.size());
  chunk(png, "IEND", nullptr, 0);
  return SNERF_IO_OK;
}

}  // namespace

extern "C" int snerf_io_version(void) { return 1; }

extern "C" long snerf_png_encode(const void* pixels, int width, int height, int channels, int bit_depth, int level, int threads, void* out,
                                 long capacity) {
  std::vector<unsigned char> png;
  const int rc = encode(pixels, width, height, channels, bit_depth, level, threads, png);
  if (rc != SNERF_IO_OK) return -(long)rc;
  if (out == nullptr || capacity < (long)png.size()) return -(long)SNERF_IO_ERR_CAPACITY;
  std::memcpy(out, png.data(), png.size());
  return (long)png.size();
}

extern "C" int snerf_png_write(const char* path, const void* pixels, int width, int height, int channels, int bit_depth, int level, int threads) {
  if (path == nullptr) return SNERF_IO_ERR_ARG;
  std::vector<unsigned char> png;
  const int rc = encode(pixels, width, height, channels, bit_depth, level, threads, png);
  if (rc != SNERF_IO_OK) return rc;
  FILE* f = std::fopen(path, "wb");
  if (f == nullptr) return SNERF_IO_ERR_FILE;
  const size_t w = std::fwrite(png.data(), 1, png.size(), f);
  const int c = std::fclose(f);
  return (w == png.size() && c == 0) ? SNERF_IO_OK : SNERF_IO_ERR_FILE;
}
