// wave.h -- 64-lane wavefront primitives for gfx950 (CDNA4).  Device code only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

namespace pw {

constexpr int WAVE = 64;

__device__ __forceinline__ int lane_id() { return (int)__lane_id(); }

// v_readlane_b32 with a wave-uniform lane index
__device__ __forceinline__ uint32_t readlane_u32(uint32_t v, int l) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, l);
}
__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int l) {
    uint32_t lo = readlane_u32((uint32_t)v, l);
    uint32_t hi = readlane_u32((uint32_t)(v >> 32), l);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ float readlane_f32(float v, int l) {
    return __uint_as_float(readlane_u32(__float_as_uint(v), l));
}
__device__ __forceinline__ double readlane_f64(double v, int l) {
    return __longlong_as_double((long long)readlane_u64((uint64_t)__double_as_longlong(v), l));
}
__device__ __forceinline__ uint32_t readfirst_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ uint64_t readfirst_u64(uint64_t v) {
    uint32_t lo = readfirst_u32((uint32_t)v);
    uint32_t hi = readfirst_u32((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

template <typename U> __device__ __forceinline__ U readlane_uint(U v, int l);
template <> __device__ __forceinline__ uint32_t readlane_uint<uint32_t>(uint32_t v, int l) { return readlane_u32(v, l); }
template <> __device__ __forceinline__ uint64_t readlane_uint<uint64_t>(uint64_t v, int l) { return readlane_u64(v, l); }

template <typename T> __device__ __forceinline__ T readlane_fp(T v, int l);
template <> __device__ __forceinline__ float readlane_fp<float>(float v, int l) { return readlane_f32(v, l); }
template <> __device__ __forceinline__ double readlane_fp<double>(double v, int l) { return readlane_f64(v, l); }

// value of lane (lane - delta); lanes < delta receive their own value
template <typename U> __device__ __forceinline__ U shfl_up_uint(U v, int delta);
template <> __device__ __forceinline__ uint32_t shfl_up_uint<uint32_t>(uint32_t v, int delta) {
    return (uint32_t)__shfl_up((int)v, (unsigned)delta, WAVE);
}
template <> __device__ __forceinline__ uint64_t shfl_up_uint<uint64_t>(uint64_t v, int delta) {
    return (uint64_t)__shfl_up((long long)v, (unsigned)delta, WAVE);
}

__device__ __forceinline__ uint64_t ballot(bool p) { return __ballot(p); }

// ---- data-parallel-primitive (DPP) sums: a lane adds the value of another lane as an operand modifier of the add itself --
// no LDS crossbar trip (ds_bpermute), no address register, no lgkmcnt wait.  Inclusive scan over the 64 lanes: row_shr 1, 2,
// 4, 8 inside each row of 16 lanes, then lane 15 of rows 0 / 2 into rows 1 / 3 (row_bcast:15) and lane 31 into rows 2, 3
// (row_bcast:31); lanes without a source add 0.  The total is what lane 63 holds.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_add_u32(uint32_t v) {
    return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
    v = dpp_add_u32<0x111, 0xF>(v);
    v = dpp_add_u32<0x112, 0xF>(v);
    v = dpp_add_u32<0x114, 0xF>(v);
    v = dpp_add_u32<0x118, 0xF>(v);
    v = dpp_add_u32<0x142, 0xA>(v);
    v = dpp_add_u32<0x143, 0xC>(v);
    return v;
}
// the same scan over float64 values (two DPP moves + one add per level; lanes without a source add +0.0: exact)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add_f64(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, CTRL, ROW_MASK, 0xF, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b >> 32), CTRL, ROW_MASK, 0xF, false);
    return v + __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double wave_incl_scan_f64(double v) {
    v = dpp_add_f64<0x111, 0xF>(v);
    v = dpp_add_f64<0x112, 0xF>(v);
    v = dpp_add_f64<0x114, 0xF>(v);
    v = dpp_add_f64<0x118, 0xF>(v);
    v = dpp_add_f64<0x142, 0xA>(v);
    v = dpp_add_f64<0x143, 0xC>(v);
    return v;
}
// the value of the lane a DPP control names (0 for lanes without a source): building block of scans whose operator is not a
// plain add (saturating sums, parity-function composition: walk_sparse.hip.h)
template <int CTRL, int ROW_MASK> __device__ __forceinline__ uint32_t dpp_get(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, false);
}
template <int CTRL, int ROW_MASK> __device__ __forceinline__ uint64_t dpp_get(uint64_t v) {
    const uint32_t lo = dpp_get<CTRL, ROW_MASK>((uint32_t)v), hi = dpp_get<CTRL, ROW_MASK>((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}
// the six levels of the 64-lane inclusive scan: STEP(control, row mask)
#define PW_DPP_SCAN_LEVELS(STEP) STEP(0x111, 0xF) STEP(0x112, 0xF) STEP(0x114, 0xF) STEP(0x118, 0xF) STEP(0x142, 0xA) STEP(0x143, 0xC)
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {   // wave-uniform result (a scalar register)
    return (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan_u32(v), WAVE - 1);
}

// orders this wave's LDS traffic (cross-lane hand-off through LDS inside one wavefront)
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ---- scalar-unit access ------------------------------------------------------------------------------
// The walk kernel carries far more wave-uniform state than the 102 SGPRs of a wave; what the allocator
// cannot keep it parks in VGPR lanes, and every v_readlane/v_writelane it then emits takes a VALU issue
// slot (a quarter of the kernel's VALU instructions before this was introduced).  Two remedies:
//   kernarg<T>(off)  re-reads a kernel argument from the kernarg segment at the point of use (s_load,
//                    volatile so that it is neither hoisted nor kept live), and
//   sptr<T>          constant-address-space view of READ-ONLY device arrays: a load through it with a
//                    wave-uniform address is an s_load (no VGPR, no readfirstlane, no VALU slot).
template <typename T> using sptr = const __attribute__((address_space(4))) T *;
template <typename T> using gptr = const __attribute__((address_space(1))) T *;
template <typename T> using gptr_mut = __attribute__((address_space(1))) T *;
template <typename T> __device__ __forceinline__ sptr<T> as_scalar(uint64_t addr) { return (sptr<T>)addr; }
template <typename T> __device__ __forceinline__ gptr<T> as_global(uint64_t addr) { return (gptr<T>)addr; }
template <typename T> __device__ __forceinline__ T kernarg(size_t off) {
    sptr<uint8_t> base = (sptr<uint8_t>)__builtin_amdgcn_kernarg_segment_ptr();
    return *(volatile const __attribute__((address_space(4))) T *)(base + off);
}

}  // namespace pw
