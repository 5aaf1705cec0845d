// Host-side mirror of src/image.zig:6-37.
#pragma once
#include <cstdint>
#include <cstdio>
#include <vector>

namespace rt {
namespace img_io {

enum class ImgType { ppm, png, jpeg };   // image.zig:6-10 (only ppm is implemented there, too)

struct Image_s {                         // image.zig:12-16
    std::vector<uint8_t> arr;            // interleaved r,g,b (the reference's ArrayList(@Vector(3,u8)))
    size_t ny = 0;
    size_t nx = 0;
};

// write_ppm_image, image.zig:18-37: "P3\n{nx} {ny}\n255\n" then "r g b\n" per pixel.
inline bool write_ppm_image(const char *fname, const uint8_t *rgb, size_t nx, size_t ny) {
    std::FILE *f = std::fopen(fname, "w");
    if (!f) return false;
    std::fprintf(f, "P3\n%zu %zu\n255\n", nx, ny);
    for (size_t k = 0; k < nx * ny; ++k)
        std::fprintf(f, "%u %u %u\n", (unsigned)rgb[3 * k], (unsigned)rgb[3 * k + 1], (unsigned)rgb[3 * k + 2]);
    return std::fclose(f) == 0;
}
inline bool write_ppm_image(const char *fname, const Image_s &img) {
    return write_ppm_image(fname, img.arr.data(), img.nx, img.ny);
}

}  // namespace img_io
}  // namespace rt
