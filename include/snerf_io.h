/* C-ABI of libsnerf_io.so: the host half of the S-NeRF++ frame writer.  Replaces the
 * PIL.Image.fromarray(...).save(path) calls of s-nerfpp/zipnerf/random_render_waymo_seq.py:214-227 and internal/utils.py:111-116
 * (save_img_u8) for the buffers snerf_frame_quantize (include/snerf_hip.h) produces.  Host-only code (g++ + zlib), no GPU runtime. */
#ifndef SNERF_IO_H
#define SNERF_IO_H
#ifdef __cplusplus
extern "C" {
#endif

#define SNERF_IO_OK 0
#define SNERF_IO_ERR_ARG 1
#define SNERF_IO_ERR_ZLIB 2
#define SNERF_IO_ERR_FILE 3
#define SNERF_IO_ERR_CAPACITY 4

int snerf_io_version(void);
/* Write `pixels` (row-major [height, width, channels], channels 1..4 = gray / gray+alpha / RGB / RGBA; bit_depth 8 = uint8 samples,
 * 16 = host-endian uint16 samples, stored big-endian as PNG requires) as a PNG file.  level = zlib level 0..9; threads = number of
 * row strips compressed concurrently (each strip is an independent deflate block run, concatenated into one zlib stream). */
int snerf_png_write(const char* path, const void* pixels, int width, int height, int channels, int bit_depth, int level, int threads);
/* The same encoder into a caller buffer: returns the number of bytes written, or -status (-SNERF_IO_ERR_CAPACITY if `capacity` is too
 * small; width*height*channels*(bit_depth/8) + height + 4096 always suffices). */
long snerf_png_encode(const void* pixels, int width, int height, int channels, int bit_depth, int level, int threads, void* out, long capacity);

#ifdef __cplusplus
}
#endif
#endif
