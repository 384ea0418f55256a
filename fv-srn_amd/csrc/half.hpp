// IEEE binary16 <-> binary32 conversion on the host (round-to-nearest-even), bit exact with
// __float2half / __half2float used by the reference host code
// (renderer/volume_interpolation_network.cpp:154,914,918).
#pragma once
#include <cstdint>
#include <cstring>

namespace fvsrn {

inline uint16_t float_to_half_bits(float f) {
    uint32_t x;
    std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) {  // inf / nan
        return static_cast<uint16_t>(sign | 0x7c00u | (x > 0x7f800000u ? (0x200u | ((x >> 13) & 0x3ffu)) : 0u));
    }
    if (x >= 0x477ff000u) {  // rounds to >= 65520 -> inf
        return static_cast<uint16_t>(sign | 0x7c00u);
    }
    if (x < 0x38800000u) {  // subnormal half or zero
        if (x < 0x33000000u) return static_cast<uint16_t>(sign);  // < 2^-25 -> 0
        const int e = static_cast<int>(x >> 23);                   // biased exponent
        const uint32_t m = (x & 0x7fffffu) | 0x800000u;
        const int shift = 126 - e;  // 14..24
        uint32_t r = m >> shift;
        const uint32_t rem = m & ((1u << shift) - 1u);
        const uint32_t halfway = 1u << (shift - 1);
        if (rem > halfway || (rem == halfway && (r & 1u))) ++r;
        return static_cast<uint16_t>(sign | r);
    }
    uint32_t r = (x - 0x38000000u) >> 13;  // rebias 127 -> 15
    const uint32_t rem = x & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) ++r;
    return static_cast<uint16_t>(sign | r);
}

inline float half_bits_to_float(uint16_t h) {
    const uint32_t sign = (static_cast<uint32_t>(h) & 0x8000u) << 16;
    const uint32_t e = (h >> 10) & 0x1fu;
    uint32_t m = h & 0x3ffu;
    uint32_t x;
    if (e == 0) {
        if (m == 0) {
            x = sign;
        } else {  // subnormal
            int k = 0;
            while (!(m & 0x400u)) { m <<= 1; ++k; }
            m &= 0x3ffu;
            x = sign | (static_cast<uint32_t>(113 - k) << 23) | (m << 13);
        }
    } else if (e == 31) {
        x = sign | 0x7f800000u | (m << 13);
    } else {
        x = sign | ((e + 112u) << 23) | (m << 13);
    }
    float f;
    std::memcpy(&f, &x, 4);
    return f;
}

}  // namespace fvsrn
