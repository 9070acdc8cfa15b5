// dev_common.h — device helpers shared by the generic scan and grid kernels.
#pragma once
#include "pcq_internal.h"

// The reference (Rust) never contracts a*b+c (last.rs:156-160, grid_sampling.rs:51-95); hipcc's
// default is -ffp-contract=fast.  The build passes -ffp-contract=off and this pragma pins it.
#pragma clang fp contract(off)

namespace pcqdev {

constexpr int BLOCK = 256;
constexpr int WAVES = BLOCK / 64;
constexpr int ITEMS = 8;                       // points per thread in the tiled (order-preserving) kernels
constexpr int TILE = BLOCK * ITEMS;            // points per block

__device__ __forceinline__ int32_t ld_i32(const uint8_t *p) {
    if (((uintptr_t)p & 3) == 0) return *reinterpret_cast<const int32_t *>(p);
    return (int32_t)((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24));
}
__device__ __forceinline__ uint16_t ld_u16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }

struct RawPoint {
    int32_t x, y, z;
};

struct __attribute__((packed, aligned(4))) PackedXYZ {
    int32_t x, y, z;
};

__device__ __forceinline__ RawPoint ld_xyz(const DevCols &c, uint64_t i) {
    const uint8_t *p = c.xyz + i * c.xyz_stride;
    RawPoint r;
    if (((uintptr_t)p & 3) == 0) {  // one global_load_dwordx3 (LAST blocks and even-pitch LAS records)
        const PackedXYZ v = *reinterpret_cast<const PackedXYZ *>(p);
        r.x = v.x;
        r.y = v.y;
        r.z = v.z;
    } else {
        r.x = ld_i32(p);
        r.y = ld_i32(p + 4);
        r.z = ld_i32(p + 8);
    }
    return r;
}

// The same for passes that read the positions once and keep nothing (the emit's two passes, the grid collector's pass 0):
// loaded with the non-temporal hint they do not displace each other's lines on the way (scan_count.hip: 6.4 -> 7.1 TB/s
// for the same access pattern).
typedef int32_t i32x3_a4 __attribute__((ext_vector_type(3), aligned(4)));
__device__ __forceinline__ RawPoint ld_xyz_stream(const DevCols &c, uint64_t i) {
    const uint8_t *p = c.xyz + i * c.xyz_stride;
    if (((uintptr_t)p & 3) != 0) return ld_xyz(c, i);
    const i32x3_a4 v = __builtin_nontemporal_load(reinterpret_cast<const i32x3_a4 *>(p));
    RawPoint r;
    r.x = v.x, r.y = v.y, r.z = v.z;
    return r;
}

// The predicate of last.rs:122-135 (bounds) / last.rs:259-262 (class).  For bounds `rp` is loaded.
__device__ __forceinline__ bool eval_pred(const DevCols &c, const DevPred &pr, uint64_t i, RawPoint &rp,
                                          bool &have_xyz) {
    if (pr.kind == PCQ_PRED_BOUNDS) {
        if (pr.empty) return false;
        rp = ld_xyz(c, i);
        have_xyz = true;
        return ((uint32_t)(rp.x - pr.lo[0]) <= pr.width[0]) & ((uint32_t)(rp.y - pr.lo[1]) <= pr.width[1]) &
               ((uint32_t)(rp.z - pr.lo[2]) <= pr.width[2]);
    }
    if (pr.kind == PCQ_PRED_BOUNDS_F64) {  // lazer.rs:65-69 on world = offset + scale * x (lazer_reader.rs:600-607)
        rp = ld_xyz(c, i);
        have_xyz = true;
        const double wx = c.offset[0] + c.scale[0] * (double)rp.x, wy = c.offset[1] + c.scale[1] * (double)rp.y,
                     wz = c.offset[2] + c.scale[2] * (double)rp.z;
        // AABB::contains rejects on `p < min || p > max` per axis [recalled, pasture-core 0.1.0 math/bounds.rs],
        // so a NaN coordinate (NaN scale/offset in the header) is NOT rejected
        return !((wx < pr.wmin[0]) | (wy < pr.wmin[1]) | (wz < pr.wmin[2]) | (wx > pr.wmax[0]) | (wy > pr.wmax[1]) |
                 (wz > pr.wmax[2]));
    }
    have_xyz = false;
    return (uint32_t)c.cls[i * c.cls_stride] == pr.cls;
}

// The same with the predicate kind fixed at compile time, so that unrolled callers get straight-line code
// (loads of several points issued together instead of one branch diamond per point).
template <int KIND>
__device__ __forceinline__ bool eval_pred_kind(const DevCols &c, const DevPred &pr, uint64_t i) {
    if (KIND == PCQ_PRED_CLASS) return (uint32_t)c.cls[i * c.cls_stride] == pr.cls;
    const RawPoint rp = ld_xyz(c, i);
    if (KIND == PCQ_PRED_BOUNDS)
        return (pr.empty == 0) & ((uint32_t)(rp.x - pr.lo[0]) <= pr.width[0]) & ((uint32_t)(rp.y - pr.lo[1]) <= pr.width[1]) &
               ((uint32_t)(rp.z - pr.lo[2]) <= pr.width[2]);
    const double wx = c.offset[0] + c.scale[0] * (double)rp.x, wy = c.offset[1] + c.scale[1] * (double)rp.y,
                 wz = c.offset[2] + c.scale[2] * (double)rp.z;
    return !((wx < pr.wmin[0]) | (wy < pr.wmin[1]) | (wz < pr.wmin[2]) | (wx > pr.wmax[0]) | (wy > pr.wmax[1]) | (wz > pr.wmax[2]));
}

// Same, keeping the loaded position (bounds kinds) for the record that a match needs.
template <int KIND>
__device__ __forceinline__ bool eval_pred_kind(const DevCols &c, const DevPred &pr, uint64_t i, RawPoint &rp) {
    if (KIND == PCQ_PRED_CLASS) return (uint32_t)c.cls[i * c.cls_stride] == pr.cls;
    rp = ld_xyz(c, i);
    if (KIND == PCQ_PRED_BOUNDS)
        return (pr.empty == 0) & ((uint32_t)(rp.x - pr.lo[0]) <= pr.width[0]) & ((uint32_t)(rp.y - pr.lo[1]) <= pr.width[1]) &
               ((uint32_t)(rp.z - pr.lo[2]) <= pr.width[2]);
    const double wx = c.offset[0] + c.scale[0] * (double)rp.x, wy = c.offset[1] + c.scale[1] * (double)rp.y,
                 wz = c.offset[2] + c.scale[2] * (double)rp.z;
    return !((wx < pr.wmin[0]) | (wy < pr.wmin[1]) | (wz < pr.wmin[2]) | (wx > pr.wmax[0]) | (wy > pr.wmax[1]) | (wz > pr.wmax[2]));
}

// last.rs:156-160 — (i as f64 * scale) + offset, two roundings.
__device__ __forceinline__ double world(int32_t v, double scale, double offset) {
    const double m = (double)v * scale;
    return m + offset;
}

// Rust `f64 as u64`: truncate, saturate, NaN -> 0.
__device__ __forceinline__ uint64_t f64_as_u64(double v) {
    if (!(v > 0.0)) return 0;
    if (v >= 18446744073709551616.0) return ~0ull;
    return (uint64_t)v;
}

// Builds the 31-byte result record of readers/src/lib.rs:10-19 for matched point i.
__device__ __forceinline__ void make_point(const DevCols &c, uint64_t i, const RawPoint &rp, pcq_point &out) {
    out.x = world(rp.x, c.scale[0], c.offset[0]);
    out.y = world(rp.y, c.scale[1], c.offset[1]);
    out.z = world(rp.z, c.scale[2], c.offset[2]);
    if (c.rgb) {  // last.rs:145-153
        const uint8_t *q = c.rgb + i * c.rgb_stride;
        out.r = ld_u16(q);
        out.g = ld_u16(q + 2);
        out.b = ld_u16(q + 4);
    } else {
        out.r = out.g = out.b = 0;
    }
    out.classification = c.cls ? c.cls[i * c.cls_stride] : 0;  // last.rs:138-142
}

__device__ __forceinline__ void store_point31(uint8_t *dst, const pcq_point &p) {
    const uint8_t *s = reinterpret_cast<const uint8_t *>(&p);
#pragma unroll
    for (int k = 0; k < 31; k++) dst[k] = s[k];
}

// The same record into a 32-byte aligned 32-byte slot as two 16-byte stores (byte 31 = 0).
__device__ __forceinline__ void store_point_slot32(uint8_t *dst32, const pcq_point &p) {
    const uint64_t bx = (uint64_t)__double_as_longlong(p.x), by = (uint64_t)__double_as_longlong(p.y),
                   bz = (uint64_t)__double_as_longlong(p.z);
    uint4 a, b;
    a.x = (uint32_t)bx, a.y = (uint32_t)(bx >> 32), a.z = (uint32_t)by, a.w = (uint32_t)(by >> 32);
    b.x = (uint32_t)bz, b.y = (uint32_t)(bz >> 32);
    b.z = (uint32_t)p.r | ((uint32_t)p.g << 16);
    b.w = (uint32_t)p.b | ((uint32_t)p.classification << 16);
    uint4 *d = reinterpret_cast<uint4 *>(dst32);
    d[0] = a;
    d[1] = b;
}

__device__ __forceinline__ uint64_t hash64(uint64_t k) {
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdull;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ull;
    k ^= k >> 33;
    return k;
}

}  // namespace pcqdev
