// vecmath.hpp -- host-side Vec3 / RNG used by scene construction.
//
// The host only BUILDS scenes and cameras (presets.rs, camera.rs:22-54,
// perlin.rs:15-51, bvh.rs:64-94); the per-ray arithmetic lives in the HIP
// kernels. Built with -ffp-contract=off so values match the reference's
// separately rounded f32 operations.
#pragma once
#include <cmath>
#include <cstdint>

namespace pt {

struct Vec3 {
    float x = 0.f, y = 0.f, z = 0.f;
    constexpr Vec3() = default;
    constexpr Vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    static constexpr Vec3 splat(float s) { return Vec3(s, s, s); }
    float operator[](int axis) const { return axis == 0 ? x : (axis == 1 ? y : z); }
};

inline Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator*(Vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline Vec3 operator*(float s, Vec3 a) { return {s * a.x, s * a.y, s * a.z}; }
// glam scalar Vec3::dot
inline float dot(Vec3 a, Vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline float length(Vec3 a) { return std::sqrt(dot(a, a)); }
// glam 0.20 scalar Vec3::normalize = v * (1 / length)
inline Vec3 normalize(Vec3 a) { return a * (1.0f / length(a)); }
inline Vec3 cross(Vec3 a, Vec3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
inline Vec3 vmin(Vec3 a, Vec3 b) { return {a.x < b.x ? a.x : b.x, a.y < b.y ? a.y : b.y, a.z < b.z ? a.z : b.z}; }
inline Vec3 vmax(Vec3 a, Vec3 b) { return {a.x > b.x ? a.x : b.x, a.y > b.y ? a.y : b.y, a.z > b.z ? a.z : b.z}; }

// rand_xoshiro 0.6 Xoshiro256Plus + rand 0.8 sampling (params.rs:21-27 seeds it with 0)
class Xoshiro256Plus {
public:
    static Xoshiro256Plus seed_from_u64(uint64_t seed) {
        Xoshiro256Plus r;
        uint64_t x = seed;
        for (auto &w : r.s_) {  // SplitMix64 stream
            x += 0x9e3779b97f4a7c15ULL;
            uint64_t z = x;
            z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
            z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
            w = z ^ (z >> 31);
        }
        return r;
    }
    uint64_t next_u64() {
        const uint64_t result = s_[0] + s_[3];
        const uint64_t t = s_[1] << 17;
        s_[2] ^= s_[0];
        s_[3] ^= s_[1];
        s_[1] ^= s_[2];
        s_[0] ^= s_[3];
        s_[2] ^= t;
        s_[3] = (s_[3] << 45) | (s_[3] >> 19);
        ++draws_;
        return result;
    }
    uint32_t next_u32() { return static_cast<uint32_t>(next_u64() >> 32); }
    // rng.gen::<f32>(): 24 random bits scaled by 2^-24
    float gen_f32() { return static_cast<float>(next_u32() >> 8) * (1.0f / 16777216.0f); }
    // rng.gen_range(low..high) for i32 (UniformInt::sample_single, widening-multiply rejection)
    int32_t gen_range(int32_t low, int32_t high) {
        const uint32_t range = static_cast<uint32_t>(high - low);
        const uint32_t zone = (range << __builtin_clz(range)) - 1u;
        for (;;) {
            const uint64_t m = static_cast<uint64_t>(next_u32()) * range;
            if (static_cast<uint32_t>(m) <= zone) return low + static_cast<int32_t>(m >> 32);
        }
    }
    uint64_t draws() const { return draws_; }

private:
    uint64_t s_[4] = {0, 0, 0, 0};
    uint64_t draws_ = 0;
};

}  // namespace pt
