// offline.cpp -- offline.rs:16-60 harness: build, time Scene::update, print the
// reference's throughput line, tone-map and write output.png.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <stdexcept>

#include "host.hpp"

namespace pt {

// math.rs:36-48. Rust `as u8` saturates (NaN -> 0).
void linear_to_srgb(const float rgb[3], uint8_t out[3]) {
    for (int i = 0; i < 3; ++i) {
        const float c = rgb[i] > 0.0f ? rgb[i] : 0.0f;
        float s = 1.055f * std::pow(c, 0.41666666f) - 0.055f;
        s = s > 0.0f ? s : 0.0f;
        s = s < 1.0f ? s : 1.0f;
        const float v = s * 255.99f;
        out[i] = !(v > 0.0f) ? 0 : (v >= 255.0f ? 255 : static_cast<uint8_t>(v));
    }
}

namespace {

struct Crc32 {
    uint32_t table[256];
    Crc32() {
        for (uint32_t n = 0; n < 256; ++n) {
            uint32_t c = n;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? (0xedb88320u ^ (c >> 1)) : (c >> 1);
            table[n] = c;
        }
    }
    uint32_t run(uint32_t crc, const uint8_t *p, size_t n) const {
        for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xffu] ^ (crc >> 8);
        return crc;
    }
};

void be32(std::vector<uint8_t> &v, uint32_t x) {
    v.push_back(uint8_t(x >> 24));
    v.push_back(uint8_t(x >> 16));
    v.push_back(uint8_t(x >> 8));
    v.push_back(uint8_t(x));
}

void chunk(std::vector<uint8_t> &png, const char tag[4], const std::vector<uint8_t> &data) {
    static const Crc32 crc;
    be32(png, static_cast<uint32_t>(data.size()));
    const size_t start = png.size();
    png.insert(png.end(), tag, tag + 4);
    png.insert(png.end(), data.begin(), data.end());
    be32(png, crc.run(0xffffffffu, png.data() + start, png.size() - start) ^ 0xffffffffu);
}

}  // namespace

// offline.rs:43-59: rows are written top-down = buffer rows reversed; RGB8. The PNG uses
// stored (uncompressed) deflate blocks: same pixels as image::save_buffer, larger file.
bool save_png(const std::string &path, const float *buffer, uint32_t width, uint32_t height) {
    std::vector<uint8_t> raw;
    raw.reserve((size_t)height * (1 + (size_t)width * 3));
    for (uint32_t row = height; row-- > 0;) {
        raw.push_back(0);  // filter: none
        for (uint32_t x = 0; x < width; ++x) {
            uint8_t px[3];
            linear_to_srgb(buffer + 3 * ((size_t)row * width + x), px);
            raw.insert(raw.end(), px, px + 3);
        }
    }
    std::vector<uint8_t> z;
    z.push_back(0x78);
    z.push_back(0x01);
    uint32_t a = 1, b = 0;  // adler32
    size_t pos = 0;
    do {
        const size_t n = std::min<size_t>(65535, raw.size() - pos);
        z.push_back(pos + n == raw.size() ? 1 : 0);
        z.push_back(uint8_t(n));
        z.push_back(uint8_t(n >> 8));
        z.push_back(uint8_t(~n));
        z.push_back(uint8_t((~n) >> 8));
        for (size_t i = 0; i < n; ++i) {
            a = (a + raw[pos + i]) % 65521u;
            b = (b + a) % 65521u;
        }
        z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
        pos += n;
    } while (pos < raw.size());
    be32(z, (b << 16) | a);

    std::vector<uint8_t> png = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    std::vector<uint8_t> ihdr;
    be32(ihdr, width);
    be32(ihdr, height);
    ihdr.insert(ihdr.end(), {8, 2, 0, 0, 0});
    chunk(png, "IHDR", ihdr);
    chunk(png, "IDAT", z);
    chunk(png, "IEND", {});
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return false;
    const bool ok = fwrite(png.data(), 1, png.size(), f) == png.size();
    return fclose(f) == 0 && ok;
}

// offline.rs:16-60 (+ `-F frames`: glium_window.rs:94-133 progressive accumulation, headless)
int render_offline(const std::string &preset, const Params &params, int device, const std::string &output,
                   uint32_t frames) {
    try {
        Xoshiro256Plus rng = params.new_rng();
        Storage storage(rng);
        auto built = presets::from_name(preset, params, rng, storage);
        if (!built) {
            fprintf(stderr, "unrecognised preset\n");  // offline.rs:21 .expect("unrecognised preset")
            return 2;
        }
        auto scene = Scene::new_scene(params, rng, storage, built->hitables, built->sky, device);
        std::vector<float> rgb_buffer((size_t)params.width * params.height * 3, 0.0f);

        for (uint32_t frame_num = 0; frame_num < (frames ? frames : 1); ++frame_num) {
            const auto start_time = std::chrono::steady_clock::now();
            const size_t ray_count = scene->update(params, built->camera, frame_num, rgb_buffer.data());
            const double elapsed_secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - start_time).count();
            // offline.rs:36-41
            printf("%.2fsecs %zurays %.2fMrays/s\n", elapsed_secs, ray_count, (double)ray_count / 1000000.0 / elapsed_secs);
        }
        if (!save_png(output, rgb_buffer.data(), params.width, params.height)) {
            fprintf(stderr, "Failed to save output image\n");  // offline.rs:59
            return 3;
        }
        return 0;
    } catch (const std::exception &e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
}

}  // namespace pt
