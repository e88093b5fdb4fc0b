/*
 * svs_oracle.c -- TEST INFRASTRUCTURE ONLY (see svs_oracle.h).
 *
 * Scalar C restatement of the reference CPU path.  Floating-point results are meant to be bit-identical to
 * the reference's AVX-512 kernels: every fused multiply-add is an explicit fmaf(), every other operation a
 * separately rounded IEEE-754 binary32 operation (compile with -ffp-contract=off, FMA hardware or a correct
 * libm fmaf()).  Reference paths are relative to the reference root (include/svs/...).
 */
#include "svs_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

size_t svso_dtype_size(int dtype) {
    switch (dtype) {
        case SVSO_F32: return 4;
        case SVSO_F16: return 2;
        case SVSO_I8:
        case SVSO_U8: return 1;
        default: return 0;
    }
}

/* ------------------------------------------------------------------------------------------------------ */
/* Float16: core/distance/simd_utils.h:271-277 loads halves with _mm512_cvtph_ps (exact, denormals kept). */
float svso_half_to_float(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1Fu;
    uint32_t man = h & 0x3FFu;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else { /* subnormal half -> normal float */
            int e = -1;
            do {
                man <<= 1;
                ++e;
            } while ((man & 0x400u) == 0);
            man &= 0x3FFu;
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
        }
    } else if (exp == 31) {
        bits = sign | 0x7F800000u | (man << 13);
    } else {
        bits = sign | ((exp + (127 - 15)) << 23) | (man << 13);
    }
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

uint16_t svso_float_to_half(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t absx = x & 0x7FFFFFFFu;
    if (absx >= 0x7F800000u) { /* inf / nan */
        return (uint16_t)(sign | 0x7C00u | ((absx > 0x7F800000u) ? 0x200u : 0));
    }
    if (absx >= 0x477FF000u) { /* rounds to >= 65520 -> inf */
        return (uint16_t)(sign | 0x7C00u);
    }
    if (absx < 0x33000001u) { /* < 2^-25 (+1ulp) -> 0 */
        return (uint16_t)sign;
    }
    int32_t e = (int32_t)(absx >> 23) - 127;
    uint32_t m = (absx & 0x7FFFFFu) | 0x800000u;
    uint32_t shift;
    uint32_t hexp;
    if (e < -14) { /* subnormal half */
        shift = (uint32_t)(13 + (-14 - e));
        hexp = 0;
    } else {
        shift = 13;
        hexp = (uint32_t)(e + 15);
    }
    uint32_t q = m >> shift;
    uint32_t rem = m & ((1u << shift) - 1u);
    uint32_t half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1u))) {
        ++q;
    }
    uint32_t out;
    if (hexp == 0) {
        out = q; /* may carry into exponent 1: still correct encoding */
    } else {
        out = ((hexp - 1) << 10) + q; /* q includes the hidden bit (0x400) */
    }
    return (uint16_t)(sign | out);
}

/* Float16::operator float (lib/float16.h:45-52,103), the conversion every SCALAR piece of the reference uses -- norm() of a float16 query
 * (distance_core.h:45-58: `accum += i * i` converts both operands) and the generic_* loops: a bit manipulation "without norm/denorm" that
 * flushes float16 denormals to (signed) zero and gives the inf / NaN exponent finite values ((31 + 112) << 23 | mantissa << 13).  Pinned
 * against the reference's own class by tests/test_oracle_vs_reference.py (scalar loops, query norms).  Round 6: the oracle used the exact
 * conversion here before, which differs on exactly those inputs. */
float svso_half_to_float_scalar(uint16_t x) {
    const uint32_t e = ((uint32_t)x & 0x7C00u) >> 10;
    const uint32_t m = ((uint32_t)x & 0x03FFu) << 13;
    const uint32_t bits = (((uint32_t)x & 0x8000u) << 16) | (e != 0 ? (((e + 112u) << 23) | m) : 0u);
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

static inline float load_as_float_scalar(const void* p, int dtype, size_t i) {
    if (dtype == SVSO_F16) return svso_half_to_float_scalar(((const uint16_t*)p)[i]);
    return dtype == SVSO_F32 ? ((const float*)p)[i] : dtype == SVSO_I8 ? (float)((const int8_t*)p)[i] : (float)((const uint8_t*)p)[i];
}

static inline float load_as_float(const void* p, int dtype, size_t i) {
    switch (dtype) {
        case SVSO_F32: return ((const float*)p)[i];
        case SVSO_F16: return svso_half_to_float(((const uint16_t*)p)[i]);
        case SVSO_I8: return (float)((const int8_t*)p)[i];  /* cvtepi8_epi32 -> cvtepi32_ps */
        default: return (float)((const uint8_t*)p)[i];      /* cvtepu8_epi32 -> cvtepi32_ps */
    }
}

/* core/distance/distance_core.h:45-66: accum += i*i in element order, then std::sqrt.  The product is formed in
 * the element's promoted type (int for i8/u8; float for f32/f16).  The reference instantiates this in the
 * caller's TU (base flags -march=x86-64: no FMA), so multiply and add round separately. */
float svso_norm(const void* x, int dtype, size_t d) {
    float accum = 0.0f;
    for (size_t i = 0; i < d; ++i) {
        float sq;
        if (dtype == SVSO_I8) {
            int v = ((const int8_t*)x)[i];
            sq = (float)(v * v);
        } else if (dtype == SVSO_U8) {
            int v = ((const uint8_t*)x)[i];
            sq = (float)(v * v);
        } else {
            float v = load_as_float_scalar(x, dtype, i);
            sq = v * v;
        }
        accum = accum + sq;
    }
    return sqrtf(accum);
}

/* ------------------------------------------------------------------------------------------------------ */
/* simd::generic_simd_op (core/distance/simd_utils.h:204-252) with a 16-lane float op:
 *   L2FloatOp<16>     euclidean.h:240-259      c=a-b; acc=fma(c,c,acc)
 *   IPFloatOp<16>     inner_product.h:199-216  acc=fma(a,b,acc)
 *   CosineFloatOp<16> cosine.h:224-256         op=fma(a,b,op); norm=fma(b,b,norm)
 * reduce = _mm512_reduce_add_ps as expanded by GCC 11 avx512fintrin.h (__MM512_REDUCE_OP).              */
static inline float reduce16(const float* v) {
    float t[8], u[4];
    for (int j = 0; j < 8; ++j) t[j] = v[j + 8] + v[j];
    for (int j = 0; j < 4; ++j) u[j] = t[j + 4] + t[j];
    float w0 = u[0] + u[2];
    float w1 = u[1] + u[3];
    return w0 + w1;
}

static void float_op(int metric, const void* a, int adt, const void* b, int bdt, size_t n, float* out_main, float* out_norm) {
    float s[4][16], q[4][16];
    memset(s, 0, sizeof(s));
    memset(q, 0, sizeof(q));
    size_t i = 0;
#define SVSO_ACC(J, L, IDX)                                         \
    do {                                                            \
        float av = load_as_float(a, adt, (IDX));                    \
        float bv = load_as_float(b, bdt, (IDX));                    \
        if (metric == SVSO_L2) {                                    \
            float c = av - bv;                                      \
            s[J][L] = fmaf(c, c, s[J][L]);                          \
        } else {                                                    \
            s[J][L] = fmaf(av, bv, s[J][L]);                        \
            if (metric == SVSO_COS) q[J][L] = fmaf(bv, bv, q[J][L]); \
        }                                                           \
    } while (0)

    if (i + 64 <= n) {
        for (; i + 64 <= n; i += 64) {
            for (int j = 0; j < 4; ++j)
                for (int l = 0; l < 16; ++l) SVSO_ACC(j, l, i + 16 * (size_t)j + (size_t)l);
        }
        for (int l = 0; l < 16; ++l) {
            s[0][l] = (s[0][l] + s[1][l]) + (s[2][l] + s[3][l]);
            q[0][l] = (q[0][l] + q[1][l]) + (q[2][l] + q[3][l]);
        }
    }
    for (; i + 16 <= n; i += 16) {
        for (int l = 0; l < 16; ++l) SVSO_ACC(0, l, i + (size_t)l);
    }
    if (i < n) { /* masked tail: inactive lanes keep the accumulator (mask3_fmadd) */
        size_t rem = n - i;
        for (size_t l = 0; l < rem; ++l) SVSO_ACC(0, l, i + l);
    }
#undef SVSO_ACC
    *out_main = reduce16(s[0]);
    if (out_norm) *out_norm = reduce16(q[0]);
}

/* VNNI paths: exact int32 accumulation (euclidean.h:265-313, inner_product.h:222-267, cosine.h:261-326). */
static void int_op(const void* a, const void* b, int dtype, size_t n, int32_t* l2, int32_t* ip, int32_t* bb) {
    uint32_t sl2 = 0, sip = 0, sbb = 0; /* wrap-around like the 32-bit lanes */
    for (size_t i = 0; i < n; ++i) {
        int32_t av, bv;
        if (dtype == SVSO_I8) {
            av = ((const int8_t*)a)[i];
            bv = ((const int8_t*)b)[i];
        } else {
            av = ((const uint8_t*)a)[i];
            bv = ((const uint8_t*)b)[i];
        }
        int32_t c = av - bv;
        sl2 += (uint32_t)(c * c);
        sip += (uint32_t)(av * bv);
        sbb += (uint32_t)(bv * bv);
    }
    *l2 = (int32_t)sl2;
    *ip = (int32_t)sip;
    *bb = (int32_t)sbb;
}

static int pair_is_float_simd(int adt, int bdt) {
    /* euclidean.h:318-358 (same list in inner_product.h:272-312, cosine.h:331-382) */
    if (adt == SVSO_F32) return 1;                       /* (f32, f32|u8|i8|f16) */
    if (adt == SVSO_F16 && (bdt == SVSO_F32 || bdt == SVSO_F16)) return 1;
    return 0;
}

float svso_distance(int metric, const void* a, int adt, const void* b, int bdt, size_t d, float a_norm, int* status) {
    if (status) *status = 0;
    if ((adt == SVSO_I8 && bdt == SVSO_I8) || (adt == SVSO_U8 && bdt == SVSO_U8)) {
        int32_t l2, ip, bb;
        int_op(a, b, adt, d, &l2, &ip, &bb);
        if (metric == SVSO_L2) return (float)l2;
        if (metric == SVSO_IP) return (float)ip;
        float b_norm = sqrtf((float)bb);           /* cosine.h:284-287 */
        return (float)ip / (a_norm * b_norm);
    }
    if (pair_is_float_simd(adt, bdt)) {
        float m, nb;
        float_op(metric, a, adt, b, bdt, d, &m, &nb);
        if (metric == SVSO_COS) return m / (sqrtf(nb) * a_norm); /* cosine.h:334-335 */
        return m;
    }
    /* Everything else -- the six (query, data) pairs utils/search_index.cpp:147-162 offers that have no AVX-512 specialisation: int8 x uint8,
     * int8 x float, int8 x float16, uint8 x int8, uint8 x float, uint8 x float16 -- is answered by the reference's scalar loops generic_l2
     * (euclidean.h:206-215), generic_ip (inner_product.h:167-176), generic_cosine_similarity (cosine.h:185-200): one float accumulator in
     * element order.  They are instantiated through L2Impl<..>::compute etc. (SVS_NOINLINE, euclidean.h:217-223) in the kernel translation unit,
     * which cmake/multi-arch.cmake builds with -march=cascadelake and GCC's default -ffp-contract=fast: `result += temp * temp` is ONE fma there.
     * These headers cannot be compiled here (euclidean.h -> lib/saveload.h -> toml++).  Round 6 pins the reading by a run all the same:
     * oracle/ref_scalar_loops.cpp holds the three loop bodies, is compiled with that translation unit's flags (no -ffp-contract flag) over the
     * reference's own Float16 class, and tests/test_oracle_vs_reference.py::test_scalar_loops_follow_the_compiler compares bit for bit --
     * which is how the float16 conversion of these loops (scalar, denormals flushed: load_as_float_scalar) was found to differ from the SIMD
     * kernels'.  What stays a restatement is the loop text itself.  tests/test_contraction_choices.py holds inputs on which the other
     * reading of the contraction gives other bits. */
    {
        float result = 0.0f, accum = 0.0f;
        for (size_t i = 0; i < d; ++i) {
            const float ai = load_as_float_scalar(a, adt, i), bi = load_as_float_scalar(b, bdt, i);
            if (metric == SVSO_L2) {
                const float temp = ai - bi;
                result = fmaf(temp, temp, result);
            } else {
                if (metric == SVSO_COS) accum = fmaf(bi, bi, accum);
                result = fmaf(ai, bi, result);
            }
        }
        if (status) *status = 1; /* answered by the scalar loop */
        if (metric == SVSO_COS) return result / (a_norm * sqrtf(accum)); /* cosine.h:199 */
        return result;
    }
}

int svso_distance_batch(int metric, const void* a, int adt, size_t na, const void* b, int bdt, size_t nb, size_t d, float* out) {
    size_t sa = svso_dtype_size(adt), sb = svso_dtype_size(bdt);
    if (!sa || !sb) return -1;
    for (size_t i = 0; i < na; ++i) {
        const char* ai = (const char*)a + i * d * sa;
        float an = (metric == SVSO_COS) ? svso_norm(ai, adt, d) : 0.0f;
        for (size_t j = 0; j < nb; ++j) {
            int st;
            out[i * nb + j] = svso_distance(metric, ai, adt, (const char*)b + j * d * sb, bdt, d, an, &st);
            if (st < 0) return st;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------------ */
/* VisitedFilter<uint32_t,16>  (index/vamana/filter.h:49-130) */
#define SVSO_FILTER_N 65536u
void svso_filter_reset(uint16_t* f) { memset(f, 0xFF, SVSO_FILTER_N * sizeof(uint16_t)); }
int svso_filter_contains(const uint16_t* f, uint32_t key) { return f[key & 0xFFFFu] == (uint16_t)(key >> 16); }
int svso_filter_emplace(uint16_t* f, uint32_t key) {
    uint16_t* v = &f[key & 0xFFFFu];
    int present = (*v == (uint16_t)(key >> 16));
    *v = (uint16_t)(key >> 16);
    return present;
}

/* ------------------------------------------------------------------------------------------------------ */
/* SearchBuffer  (index/vamana/search_buffer.h:104-497) */
static inline int cmp(const svso_buffer* b, float x, float y) { return b->greater ? (x > y) : (x < y); }

int svso_buffer_init(svso_buffer* b, size_t window, size_t capacity, int greater, int visited_set) {
    if (window > capacity) return -1; /* search_buffer.h:84-93 */
    memset(b, 0, sizeof(*b));
    b->greater = greater;
    b->window = window;
    b->capacity = capacity;
    b->e = (svso_neighbor*)calloc(capacity + 1, sizeof(svso_neighbor));
    if (!b->e) return -2;
    if (visited_set) {
        b->filter = (uint16_t*)malloc(SVSO_FILTER_N * sizeof(uint16_t));
        if (!b->filter) return -2;
        svso_filter_reset(b->filter);
    }
    return 0;
}

void svso_buffer_free(svso_buffer* b) {
    free(b->e);
    free(b->filter);
    b->e = NULL;
    b->filter = NULL;
}

void svso_buffer_clear(svso_buffer* b) { /* :189-195 */
    b->size = 0;
    b->best_unvisited = 0;
    if (b->filter) svso_filter_reset(b->filter);
}

void svso_buffer_push_back(svso_buffer* b, uint32_t id, float dist) { /* :306-311 */
    if (b->size != b->capacity) {
        b->e[b->size].id = id;
        b->e[b->size].dist = dist;
        b->e[b->size].visited = 0;
        b->size++;
    }
}

/* :408 std::sort(begin(), end(), compare_).  std::sort is not stable: entries of exactly equal distance come out in an
 * order fixed by the library's algorithm, and BatchIterator re-sorts an already sorted buffer on every resumed search
 * (iterator.h:61).  To stay id-exact through ties this restates libstdc++'s algorithm (GCC 11 bits/stl_algo.h:
 * __sort = __introsort_loop(depth 2*lg n; threshold 16; median of {first+1, mid, last-1} swapped to first; unguarded
 * partition of (first+1, last) around *first) followed by __final_insertion_sort).  The heapsort fallback at depth 0 is
 * not restated; it cannot trigger on sorted input (the median is the middle value, so ranges halve) and is reported
 * through svso_sort_depth_exhausted for anything else. */
static int g_sort_depth_exhausted = 0;
int svso_sort_depth_exhausted(void) { return g_sort_depth_exhausted; }

static inline int nb_cmp(const svso_buffer* b, const svso_neighbor* x, const svso_neighbor* y) { return cmp(b, x->dist, y->dist); }
static inline void nb_swap(svso_neighbor* x, svso_neighbor* y) { svso_neighbor t = *x; *x = *y; *y = t; }

static void unguarded_linear_insert(const svso_buffer* b, svso_neighbor* a, size_t last) {
    svso_neighbor val = a[last];
    size_t next = last - 1;
    while (nb_cmp(b, &val, &a[next])) {
        a[last] = a[next];
        last = next;
        --next; /* never read below the guard: a smaller-or-equal element stops the loop first */
    }
    a[last] = val;
}

static void insertion_sort(const svso_buffer* b, svso_neighbor* a, size_t first, size_t last) {
    if (first == last) return;
    for (size_t i = first + 1; i != last; ++i) {
        if (nb_cmp(b, &a[i], &a[first])) {
            svso_neighbor val = a[i];
            memmove(&a[first + 1], &a[first], (i - first) * sizeof(svso_neighbor));
            a[first] = val;
        } else {
            unguarded_linear_insert(b, a, i);
        }
    }
}

void svso_buffer_sort(svso_buffer* b) {
    svso_neighbor* a = b->e;
    const size_t n = b->size;
    if (n == 0) return;
    struct { size_t first, last; unsigned depth; } stack[128];
    int sp = 0;
    unsigned lg = 0;
    for (size_t t = n; t > 1; t >>= 1) ++lg;
    stack[sp].first = 0; stack[sp].last = n; stack[sp].depth = 2 * lg; ++sp;
    while (sp > 0) {
        --sp;
        size_t first = stack[sp].first, last = stack[sp].last;
        unsigned depth = stack[sp].depth;
        while (last - first > 16) {
            if (depth == 0) { g_sort_depth_exhausted = 1; break; }
            --depth;
            /* __move_median_to_first(first, first + 1, mid, last - 1) */
            svso_neighbor *r = &a[first], *x = &a[first + 1], *y = &a[first + (last - first) / 2], *z = &a[last - 1];
            if (nb_cmp(b, x, y)) {
                if (nb_cmp(b, y, z)) nb_swap(r, y);
                else if (nb_cmp(b, x, z)) nb_swap(r, z);
                else nb_swap(r, x);
            } else if (nb_cmp(b, x, z)) nb_swap(r, x);
            else if (nb_cmp(b, y, z)) nb_swap(r, z);
            else nb_swap(r, y);
            /* __unguarded_partition(first + 1, last, pivot = first) */
            size_t lo = first + 1, hi = last;
            for (;;) {
                while (nb_cmp(b, &a[lo], &a[first])) ++lo;
                --hi;
                while (nb_cmp(b, &a[first], &a[hi])) --hi;
                if (!(lo < hi)) break;
                nb_swap(&a[lo], &a[hi]);
                ++lo;
            }
            /* recurse on [cut, last), continue with [first, cut) */
            if (sp < 127) { stack[sp].first = lo; stack[sp].last = last; stack[sp].depth = depth; ++sp; }
            last = lo;
        }
    }
    /* __final_insertion_sort */
    if (n > 16) {
        insertion_sort(b, a, 0, 16);
        for (size_t i = 16; i != n; ++i) unguarded_linear_insert(b, a, i);
    } else {
        insertion_sort(b, a, 0, n);
    }
}

int svso_buffer_done(const svso_buffer* b) { /* :280 */
    size_t upper = b->size < b->window ? b->size : b->window;
    return b->best_unvisited == upper;
}

svso_neighbor svso_buffer_next(svso_buffer* b) { /* :294-304 */
    svso_neighbor* node = &b->e[b->best_unvisited];
    node->visited = 1;
    svso_neighbor out = *node;
    size_t upper = b->size < b->window ? b->size : b->window;
    while (++b->best_unvisited != upper && b->e[b->best_unvisited].visited) {}
    return out;
}

size_t svso_buffer_insert(svso_buffer* b, uint32_t id, float dist) { /* :342-403 */
    /* can_skip */
    if (b->size == b->capacity && (b->capacity == 0 || cmp(b, b->e[b->size - 1].dist, dist))) {
        return b->size;
    }
    /* lower_bound with predicate !cmp(dist, other): first position whose element compares strictly after */
    size_t lo = 0, hi = b->size;
    while (lo < hi) {
        size_t mid = lo + (hi - lo) / 2;
        if (!cmp(b, dist, b->e[mid].dist)) lo = mid + 1;
        else hi = mid;
    }
    size_t pos = lo;
    /* duplicate scan over the run of equal distances ending at pos */
    if (pos != 0) {
        size_t back = pos;
        do {
            --back;
            if (cmp(b, b->e[back].dist, dist)) break;
            if (b->e[back].id == id) return b->size + 1;
        } while (back != 0);
    }
    /* unsafe_insert: copy_backward [pos, size) -> one to the right (slot [capacity] is the spare) */
    for (size_t i = b->size; i > pos; --i) b->e[i] = b->e[i - 1];
    b->e[pos].id = id;
    b->e[pos].dist = dist;
    b->e[pos].visited = 0;
    b->size = (b->size + 1 < b->capacity) ? b->size + 1 : b->capacity;
    if (pos < b->best_unvisited) b->best_unvisited = pos;
    return pos;
}

/* ------------------------------------------------------------------------------------------------------ */
/* Scalar quantization  (quantization/scalar/scalar.h) */
static inline float sq_code_min(int dt) { return dt == SVSO_I8 ? -128.0f : 0.0f; }
static inline float sq_code_max(int dt) { return dt == SVSO_I8 ? 127.0f : 255.0f; }

void svso_sq_params(const float* data, size_t n, size_t d, int code_dtype, float* scale, float* bias) { /* :454-462 */
    float mn = 3.402823466e+38f, mx = -3.402823466e+38f;
    for (size_t i = 0; i < n * d; ++i) {
        if (data[i] < mn) mn = data[i];
        if (data[i] > mx) mx = data[i];
    }
    float MIN = sq_code_min(code_dtype), MAX = sq_code_max(code_dtype);
    *scale = (mx - mn) / (MAX - MIN);
    *bias = mn - MIN * (*scale);
}

static inline float sq_compress_one(float v, int code_dtype, float scale, float bias) { /* :37-42 */
    float r = roundf((v - bias) / scale);
    float MIN = sq_code_min(code_dtype), MAX = sq_code_max(code_dtype);
    if (r < MIN) r = MIN;
    if (r > MAX) r = MAX;
    return r;
}

void svso_sq_compress(const float* src, size_t count, int code_dtype, float scale, float bias, void* dst) {
    for (size_t i = 0; i < count; ++i) {
        float r = sq_compress_one(src[i], code_dtype, scale, bias);
        if (code_dtype == SVSO_I8) ((int8_t*)dst)[i] = (int8_t)r;
        else ((uint8_t*)dst)[i] = (uint8_t)r;
    }
}

/* Per-query state produced by fix_argument (concepts/distance.h:88-92). */
typedef struct {
    const void* q;      /* query as used by the kernel */
    int q_dtype;
    float norm;         /* cosine */
    float offset;       /* SQ inner product: bias * sum(q) */
    void* owned;        /* scratch to free */
} fixed_query;

static int fix_query(const svso_index* ix, const void* query, int q_dtype, fixed_query* fq) {
    memset(fq, 0, sizeof(*fq));
    fq->q = query;
    fq->q_dtype = q_dtype;
    if (!ix->sq_enabled) {
        if (ix->metric == SVSO_COS) fq->norm = svso_norm(query, q_dtype, ix->d); /* cosine.h:117-119 */
        return 0;
    }
    /* SQ distances take float queries (scalar.h:58-195); f16 queries are widened element by element -- detail::compress<T>(v, ..) on a Float16
     * value (scalar.h:75-82) and set_datum -> relaxed_narrow<float>(Float16) (scalar.h:123-125, core/data/simple.h:396-404) -- i.e. through
     * Float16::operator float, the scalar conversion */
    if (q_dtype != SVSO_F32 && q_dtype != SVSO_F16) return -1;
    float* qf = (float*)malloc(ix->d * sizeof(float));
    if (!qf) return -2;
    for (size_t i = 0; i < ix->d; ++i) qf[i] = load_as_float_scalar(query, q_dtype, i);
    if (ix->metric == SVSO_L2) { /* EuclideanCompressed::fix_argument :75-82 */
        void* qc = malloc(ix->d);
        if (!qc) { free(qf); return -2; }
        svso_sq_compress(qf, ix->d, ix->data_dtype, ix->sq_scale, ix->sq_bias, qc);
        free(qf);
        fq->q = qc;
        fq->q_dtype = ix->data_dtype;
        fq->owned = qc;
    } else if (ix->metric == SVSO_IP) { /* InnerProductCompressed::fix_argument :122-130; libstdc++-11 std::reduce */
        float init = 0.0f;
        size_t i = 0;
        for (; ix->d - i >= 4; i += 4) {
            float v1 = qf[i] + qf[i + 1];
            float v2 = qf[i + 2] + qf[i + 3];
            float v3 = v1 + v2;
            init = init + v3;
        }
        for (; i < ix->d; ++i) init = init + qf[i];
        fq->offset = ix->sq_bias * init;
        fq->q = qf;
        fq->q_dtype = SVSO_F32;
        fq->owned = qf;
    } else { /* CosineSimilarityCompressed::fix_argument :168-171 */
        fq->norm = svso_norm(qf, SVSO_F32, ix->d);
        fq->q = qf;
        fq->q_dtype = SVSO_F32;
        fq->owned = qf;
    }
    return 0;
}

static float index_distance(const svso_index* ix, const fixed_query* fq, uint32_t id, float* scratch) {
    const char* row = (const char*)ix->data + (size_t)id * ix->d * svso_dtype_size(ix->data_dtype);
    int st;
    if (!ix->sq_enabled) {
        return svso_distance(ix->metric, fq->q, fq->q_dtype, row, ix->data_dtype, ix->d, fq->norm, &st);
    }
    if (ix->metric == SVSO_L2) { /* scalar.h:88-94: scale^2 * L2(codes) */
        float inner = svso_distance(SVSO_L2, fq->q, fq->q_dtype, row, ix->data_dtype, ix->d, 0.0f, &st);
        return (ix->sq_scale * ix->sq_scale) * inner;
    }
    if (ix->metric == SVSO_IP) { /* scalar.h:134-141: scale*ip + offset (separately rounded) */
        float ip = svso_distance(SVSO_IP, fq->q, SVSO_F32, row, ix->data_dtype, ix->d, 0.0f, &st);
        float t = ix->sq_scale * ip;
        return t + fq->offset;
    }
    /* scalar.h:175-184: decompress (scale*float(v)+bias, separately rounded) then f32 x f32 cosine */
    for (size_t i = 0; i < ix->d; ++i) {
        float t = ix->sq_scale * load_as_float(row, ix->data_dtype, i);
        scratch[i] = t + ix->sq_bias;
    }
    return svso_distance(SVSO_COS, fq->q, SVSO_F32, scratch, SVSO_F32, ix->d, fq->norm, &st);
}

/* ------------------------------------------------------------------------------------------------------ */
/* greedy_search  (index/vamana/greedy_search.h:124-203) with EntryPointInitializer (:62-94) */
static int greedy_search_core(
    const svso_index* ix, const void* query, int q_dtype, svso_buffer* buf, svso_stats* stats,
    uint32_t* history, float* history_dist, size_t history_cap, size_t* history_len, int resume
) {
    fixed_query fq;
    int rc = fix_query(ix, query, q_dtype, &fq);
    if (rc) return rc;
    {
        int st = 0;
        if (!ix->sq_enabled) {
            (void)svso_distance(ix->metric, fq.q, fq.q_dtype, ix->data, ix->data_dtype, 0, 1.0f, &st);
            if (st < 0) { free(fq.owned); return -1; }
        }
    }
    float* scratch = NULL;
    if (ix->sq_enabled && ix->metric == SVSO_COS) {
        scratch = (float*)malloc(ix->d * sizeof(float));
        if (!scratch) { free(fq.owned); return -2; }
    }
    svso_stats local;
    memset(&local, 0, sizeof(local));
    static __thread uint32_t sim_filter[4][4096];
    if (stats) memset(sim_filter, 0, sizeof(sim_filter));
    size_t hlen = 0;

    if (!resume) { /* EntryPointInitializer (:62-94) */
        svso_buffer_clear(buf);
        for (size_t e = 0; e < ix->n_entry; ++e) {
            uint32_t id = ix->entry_points[e];
            float dd = index_distance(ix, &fq, id, scratch);
            local.distance_evals++;
            svso_buffer_push_back(buf, id, dd);
        }
    } /* else RestartInitializer (index/vamana/iterator.h:36-63): keep the contents, only sort */
    svso_buffer_sort(buf);

    while (!svso_buffer_done(buf)) {
        svso_neighbor node = svso_buffer_next(buf);
        const uint32_t* row = ix->graph + (size_t)node.id * ix->graph_stride;
        uint32_t deg = row[0];
        local.expansions++;
        local.neighbor_slots += deg;                 /* tracker.visited(node, neighbors.size()) :165 */
        if (history && hlen < history_cap) {
            history[hlen] = node.id;
            if (history_dist) history_dist[hlen] = node.dist;
        }
        hlen++;
        for (uint32_t j = 0; j < deg; ++j) {
            uint32_t id = row[1 + j];
            if (stats) { /* analysis counters (not part of the restated path) */
                for (size_t t = 0; t < buf->size; ++t) if (buf->e[t].id == id) { local.in_buffer++; break; }
                for (int fs = 0; fs < 4; ++fs) { /* exact-tag direct-mapped filters of 512..4096 entries */
                    uint32_t mask = (512u << fs) - 1u;
                    uint32_t h = (id * 2654435761u) >> 7;
                    uint32_t* slot = &sim_filter[fs][h & mask];
                    if (*slot == id + 1u) local.small_filter_hits[fs]++;
                    *slot = id + 1u;
                }
            }
            if (buf->filter && svso_filter_emplace(buf->filter, id)) continue; /* :191-193 */
            float dd = index_distance(ix, &fq, id, scratch);
            if (stats && ix->metric == SVSO_L2 && ix->data_dtype == SVSO_F32 && q_dtype == SVSO_F32 && !ix->sq_enabled && ix->d > 64 &&
                buf->size == buf->capacity) { /* analysis: would a first-64-dims partial distance already exceed the worst? */
                const float* qv = (const float*)query;
                const float* xv = (const float*)ix->data + (size_t)id * ix->d;
                float part = 0.0f;
                for (size_t t = 0; t < 64; ++t) { float c = qv[t] - xv[t]; part += c * c; }
                if (part > buf->e[buf->size - 1].dist * 1.00004f) local.partial_rejects++;
            }
            local.distance_evals++;
            svso_buffer_insert(buf, id, dd);
        }
    }
    if (stats) *stats = local;
    if (history_len) *history_len = hlen;
    free(scratch);
    free(fq.owned);
    return 0;
}

int svso_greedy_search(
    const svso_index* ix, const void* query, int q_dtype, svso_buffer* buf, svso_stats* stats,
    uint32_t* history, float* history_dist, size_t history_cap, size_t* history_len
) {
    return greedy_search_core(ix, query, q_dtype, buf, stats, history, history_dist, history_cap, history_len, 0);
}

/* ------------------------------------------------------------------------------------------------------ */
/* BatchIterator (index/vamana/iterator.h:81-339), static index (no id translation, no reranking).
 * RestartInitializer re-sorts the already sorted buffer with std::sort (:61), which permutes entries of exactly equal
 * distance: svso_buffer_sort restates libstdc++'s algorithm so that the tie order matches too. */
int svso_buffer_change_maxsize(svso_buffer* b, size_t window, size_t capacity) { /* search_buffer.h:206-211 */
    svso_neighbor* e = (svso_neighbor*)realloc(b->e, (capacity + 1) * sizeof(svso_neighbor));
    if (!e) return -2;
    b->e = e;
    b->window = window;
    b->capacity = capacity;
    if (b->size > capacity) b->size = capacity;
    return 0;
}

int svso_iterator_init(svso_iterator* it, const svso_index* ix, const void* query, int q_dtype, size_t extra_capacity) {
    memset(it, 0, sizeof(*it));
    it->ix = ix;
    it->q_dtype = q_dtype;
    size_t qb = svso_dtype_size(q_dtype) * ix->d;
    it->query = malloc(qb ? qb : 1);
    if (!it->query) return -2;
    memcpy(it->query, query, qb);
    it->extra = extra_capacity == (size_t)-1 ? 100 : extra_capacity; /* lib/preprocessor.h:177,187; iterator.h:166-170 */
    it->restart = 1;
    /* initialize_buffer (:127-131): SearchBufferConfig{0, extra}, clear */
    int rc = svso_buffer_init(&it->buf, 0, it->extra, ix->metric != SVSO_L2, 0);
    if (rc) { free(it->query); it->query = NULL; }
    return rc;
}

void svso_iterator_free(svso_iterator* it) {
    svso_buffer_free(&it->buf);
    free(it->query);
    free(it->yielded);
    memset(it, 0, sizeof(*it));
}

int svso_iterator_done(const svso_iterator* it) { /* :230 */
    return it->n_yielded == it->ix->n || it->exhausted;
}

void svso_iterator_restart_next_search(svso_iterator* it) { it->restart = 1; } /* :233 */

int svso_iterator_update(svso_iterator* it, const void* query) { /* :177-190 */
    memcpy(it->query, query, svso_dtype_size(it->q_dtype) * it->ix->d);
    int rc = svso_buffer_change_maxsize(&it->buf, 0, it->extra);
    if (rc) return rc;
    svso_buffer_clear(&it->buf);
    it->restart = 1;
    it->iteration = 0;
    it->n_yielded = 0;
    it->exhausted = 0;
    return 0;
}

/* next (:249-317) + copy_from_scratch (:104-124): returns the number of neighbours written (<= batch_size), < 0 on error */
long svso_iterator_next(svso_iterator* it, size_t batch_size, uint32_t* ids, float* dists) {
    if (svso_iterator_done(it)) return 0;
    /* increment_buffer (:135-139) */
    int rc = svso_buffer_change_maxsize(&it->buf, it->buf.window + batch_size, it->buf.capacity + batch_size);
    if (rc) return rc;
    int hard = it->restart;
    it->restart = 1;
    rc = greedy_search_core(it->ix, it->query, it->q_dtype, &it->buf, NULL, NULL, NULL, 0, NULL, !hard);
    if (rc) return rc;
    it->iteration++;
    it->restart = 0;
    size_t got = 0;
    for (size_t i = 0; i < it->buf.size && got < batch_size; ++i) {
        uint32_t id = it->buf.e[i].id;
        int seen = 0;
        for (size_t y = 0; y < it->n_yielded; ++y) if (it->yielded[y] == id) { seen = 1; break; }
        if (seen) continue;
        if (it->n_yielded == it->yielded_cap) {
            size_t nc = it->yielded_cap ? it->yielded_cap * 2 : 256;
            uint32_t* ny = (uint32_t*)realloc(it->yielded, nc * sizeof(uint32_t));
            if (!ny) return -2;
            it->yielded = ny;
            it->yielded_cap = nc;
        }
        it->yielded[it->n_yielded++] = id;
        ids[got] = id;
        dists[got] = it->buf.e[i].dist;
        got++;
    }
    if (got == 0 && batch_size > 0) it->exhausted = 1;
    return (long)got;
}

/* VamanaIndex::search batch (index/vamana/index.h:564-611) + per_thread_batch_search (extensions.h:555-592) */
int svso_search_batch(
    const svso_index* ix, const void* queries, int q_dtype, size_t nq, size_t k, size_t window, size_t capacity,
    int visited_set, uint32_t* ids, float* dists, svso_stats* per_query_stats
) {
    if (window > capacity) return -1;
    if (capacity < k) { /* index.h:590-592 change_maxsize(SearchBufferConfig{k}) */
        window = k;
        capacity = k;
    }
    int greater = ix->metric != SVSO_L2;
    svso_buffer buf;
    int rc = svso_buffer_init(&buf, window, capacity, greater, visited_set);
    if (rc) return rc;
    size_t qs = svso_dtype_size(q_dtype) * ix->d;
    for (size_t i = 0; i < nq; ++i) {
        rc = svso_greedy_search(ix, (const char*)queries + i * qs, q_dtype, &buf, per_query_stats ? &per_query_stats[i] : NULL, NULL, NULL, 0, NULL);
        if (rc) break;
        for (size_t j = 0; j < k; ++j) {
            if (j < buf.size) {
                ids[i * k + j] = buf.e[j].id;
                dists[i * k + j] = buf.e[j].dist;
            } else { /* bindings/cpp/src/vamana_index_impl.h:205-209 sentinel convention */
                ids[i * k + j] = 0xFFFFFFFFu;
                dists[i * k + j] = INFINITY; /* Unspecify<float>() */
            }
        }
    }
    svso_buffer_free(&buf);
    return rc;
}

/* ------------------------------------------------------------------------------------------------------ */
/* k_recall_at_n  (core/recall.h:47-58): |gt[:k] ∩ res[:n]| summed over queries / (k * nq) */
double svso_k_recall_at_n(const uint32_t* gt, size_t gt_stride, const uint32_t* res, size_t res_stride, size_t nq, size_t k, size_t n) {
    size_t hits = 0;
    for (size_t q = 0; q < nq; ++q) {
        const uint32_t* g = gt + q * gt_stride;
        const uint32_t* r = res + q * res_stride;
        for (size_t i = 0; i < k; ++i) {
            for (size_t j = 0; j < n; ++j) {
                if (g[i] == r[j]) { hits++; break; }
            }
        }
    }
    return (double)hits / ((double)k * (double)nq);
}

/* Exhaustive search (index/flat/flat.h:421-460 semantics): k best by (distance, then lower id). */
int svso_flat_search(const svso_index* ix, const void* queries, int q_dtype, size_t nq, size_t k, uint32_t* ids, float* dists) {
    int greater = ix->metric != SVSO_L2;
    size_t qs = svso_dtype_size(q_dtype) * ix->d;
    float* scratch = (float*)malloc(ix->d * sizeof(float));
    float* bd = (float*)malloc(k * sizeof(float));
    uint32_t* bi = (uint32_t*)malloc(k * sizeof(uint32_t));
    if (!scratch || !bd || !bi) return -2;
    for (size_t q = 0; q < nq; ++q) {
        fixed_query fq;
        int rc = fix_query(ix, (const char*)queries + q * qs, q_dtype, &fq);
        if (rc) return rc;
        size_t sz = 0;
        for (size_t i = 0; i < ix->n; ++i) {
            float dd = index_distance(ix, &fq, (uint32_t)i, scratch);
            if (sz == k && !(greater ? dd > bd[k - 1] : dd < bd[k - 1])) continue;
            size_t p = sz < k ? sz : k - 1;
            while (p > 0 && (greater ? dd > bd[p - 1] : dd < bd[p - 1])) {
                bd[p] = bd[p - 1];
                bi[p] = bi[p - 1];
                --p;
            }
            bd[p] = dd;
            bi[p] = (uint32_t)i;
            if (sz < k) sz++;
        }
        for (size_t j = 0; j < k; ++j) {
            ids[q * k + j] = j < sz ? bi[j] : 0xFFFFFFFFu;
            dists[q * k + j] = j < sz ? bd[j] : INFINITY;
        }
        free(fq.owned);
    }
    free(scratch);
    free(bd);
    free(bi);
    return 0;
}
