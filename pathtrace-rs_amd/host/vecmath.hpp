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

inline Vec3 operator*(Vec3 a, Vec3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline Vec3 operator-(Vec3 a) { return {-a.x, -a.y, -a.z}; }

// glam 0.20 Quat (only what presets.rs:383-390 uses)
struct Quat {
    float x, y, z, w;
    // Quat::from_rotation_y: (s, c) = sin_cos(angle * 0.5); (0, s, 0, c)
    static Quat from_rotation_y(float angle) { return {0.0f, std::sin(angle * 0.5f), 0.0f, std::cos(angle * 0.5f)}; }
};
// f32::to_radians
inline float to_radians(float deg) { return deg * (3.14159274101257324f / 180.0f); }

// glam 0.20 Affine3A: matrix3 columns + translation
struct Affine3A {
    Vec3 x_axis{1.f, 0.f, 0.f}, y_axis{0.f, 1.f, 0.f}, z_axis{0.f, 0.f, 1.f}, translation{0.f, 0.f, 0.f};
    // Mat3A::from_quat
    static Affine3A from_rotation_translation(Quat q, Vec3 t) {
        const float x2 = q.x + q.x, y2 = q.y + q.y, z2 = q.z + q.z;
        const float xx = q.x * x2, xy = q.x * y2, xz = q.x * z2, yy = q.y * y2, yz = q.y * z2, zz = q.z * z2;
        const float wx = q.w * x2, wy = q.w * y2, wz = q.w * z2;
        Affine3A m;
        m.x_axis = Vec3(1.0f - (yy + zz), xy + wz, xz - wy);
        m.y_axis = Vec3(xy - wz, 1.0f - (xx + zz), yz + wx);
        m.z_axis = Vec3(xz + wy, yz - wx, 1.0f - (xx + yy));
        m.translation = t;
        return m;
    }
    // ((x_axis * v.x) + (y_axis * v.y)) + (z_axis * v.z)
    Vec3 transform_vector3(Vec3 v) const { return (x_axis * v.x + y_axis * v.y) + z_axis * v.z; }
    Vec3 transform_point3(Vec3 p) const { return transform_vector3(p) + translation; }
    // Mat3A::inverse (cross products over the determinant, transposed); translation = -(inverse * t)
    Affine3A inverse() const {
        const Vec3 tmp0 = cross(y_axis, z_axis), tmp1 = cross(z_axis, x_axis), tmp2 = cross(x_axis, y_axis);
        const float inv_det = 1.0f / dot(z_axis, tmp2);
        const Vec3 c0 = tmp0 * inv_det, c1 = tmp1 * inv_det, c2 = tmp2 * inv_det;
        Affine3A r;
        r.x_axis = Vec3(c0.x, c1.x, c2.x);
        r.y_axis = Vec3(c0.y, c1.y, c2.y);
        r.z_axis = Vec3(c0.z, c1.z, c2.z);
        r.translation = Vec3(0.f, 0.f, 0.f);
        r.translation = -r.transform_vector3(translation);
        return r;
    }
};

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
