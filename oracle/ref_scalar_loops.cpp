// ref_scalar_loops.cpp -- TEST INFRASTRUCTURE ONLY (part of oracle/_ref/libsvs_ref.so; never linked into the product).
//
// What it pins.  Six (query, data) type pairs have no AVX-512 specialisation in the reference and are answered by its scalar loops
//     generic_l2                  /root/reference/include/svs/core/distance/euclidean.h:206-215
//     generic_ip                  /root/reference/include/svs/core/distance/inner_product.h:167-176
//     generic_cosine_similarity   /root/reference/include/svs/core/distance/cosine.h:185-200
// instantiated (through L2Impl<..>::compute etc., SVS_NOINLINE) inside the kernel translation unit that cmake/multi-arch.cmake:17-35 builds
// with `-march=cascadelake -mtune=cascadelake` and cmake/options.cmake:187-189 with `-O3` -- and with NO -ffp-contract flag, i.e. GCC's default
// for C++ (`fast`).  Whether `result += temp * temp` is one fma or a multiply and an add decides the bits.  Those three headers cannot be
// compiled here (euclidean.h -> lib/saveload.h -> toml++, absent from the image), so this file holds the three loop bodies as the headers write
// them -- same statements, same types, same order -- and oracle/Makefile compiles THIS translation unit with exactly the flags above (not with
// the -ffp-contract=off the rest of oracle/ uses).  tests/test_oracle_vs_reference.py::test_scalar_loops_follow_the_compiler then compares
// the C restatement (oracle/svs_oracle.c, which hard-codes fmaf) with what the compiler really made of the loops: the assumption "GCC
// contracts them" is checked by a run, not read off the build files.  What stays unpinned is only that the loop text below is the header's.
#include "svs/lib/float16.h"

#include <cmath>
#include <cstddef>
#include <cstdint>

namespace {

// float16 elements are the reference's own class, compiled from where it lies: its scalar `operator float` (lib/float16.h:45-52,103) is a bit
// manipulation that flushes denormals to zero and maps the inf / NaN exponent to finite values -- NOT the hardware conversion the SIMD kernels use
using Half = svs::Float16;

template <typename Ea, typename Eb> __attribute__((noinline)) float loop_l2(const Ea* a, const Eb* b, size_t length) {
    float result = 0;
    for (size_t i = 0; i < length; ++i) {
        auto temp = static_cast<float>(a[i]) - static_cast<float>(b[i]);
        result += temp * temp;
    }
    return result;
}

template <typename Ea, typename Eb> __attribute__((noinline)) float loop_ip(const Ea* a, const Eb* b, size_t length) {
    float result = 0;
    for (size_t i = 0; i < length; ++i) {
        result += static_cast<float>(a[i]) * static_cast<float>(b[i]);
    }
    return result;
}

template <typename Ea, typename Eb> __attribute__((noinline)) float loop_cos(const Ea* a, const Eb* b, float a_norm, size_t length) {
    float result = 0;
    float accum = 0;
    for (size_t i = 0; i < length; ++i) {
        float bi_float = static_cast<float>(b[i]);
        accum += bi_float * bi_float;
        result += static_cast<float>(a[i]) * bi_float;
    }
    return result / (a_norm * std::sqrt(accum));
}

template <typename Ea, typename Eb> float run(int metric, const void* a, const void* b, size_t d, float a_norm) {
    const Ea* pa = static_cast<const Ea*>(a);
    const Eb* pb = static_cast<const Eb*>(b);
    return metric == 0 ? loop_l2(pa, pb, d) : metric == 1 ? loop_ip(pa, pb, d) : loop_cos(pa, pb, a_norm, d);
}

} // namespace

// dtype codes as in oracle/svs_oracle.h: 0 float32, 1 float16, 2 int8, 3 uint8.  Returns NaN-free garbage for pairs it does not hold: *ok = 0.
extern "C" float svsref_scalar_loop(int metric, const void* a, int adt, const void* b, int bdt, size_t d, float a_norm, int* ok) {
    *ok = 1;
    if (adt == 2 && bdt == 3) return run<int8_t, uint8_t>(metric, a, b, d, a_norm);
    if (adt == 2 && bdt == 0) return run<int8_t, float>(metric, a, b, d, a_norm);
    if (adt == 2 && bdt == 1) return run<int8_t, Half>(metric, a, b, d, a_norm);
    if (adt == 3 && bdt == 2) return run<uint8_t, int8_t>(metric, a, b, d, a_norm);
    if (adt == 3 && bdt == 0) return run<uint8_t, float>(metric, a, b, d, a_norm);
    if (adt == 3 && bdt == 1) return run<uint8_t, Half>(metric, a, b, d, a_norm);
    *ok = 0;
    return 0.0f;
}
