// Shared device helpers for the gfx950 (MI355X, CDNA4) kernels of libvita_hip.so.
// gfx950 only: wave = 64 lanes, native __bf16 conversions (v_cvt_pk_bf16_f32, RNE).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/vita_hip.h"
#include <atomic>

// Function attributes (dynamic LDS size) are per device: run `f` once per device ordinal, not once per process.
// Two threads racing on the same device may both run it (idempotent); nobody launches before it has run.
template <class F>
inline void vita_device_once(std::atomic<unsigned long long>& done, F&& f) {
  int d = 0;
  (void)hipGetDevice(&d);
  const unsigned long long bit = 1ull << (d & 63);
  if (!(done.load(std::memory_order_acquire) & bit)) {
    f();
    done.fetch_or(bit, std::memory_order_release);
  }
}

typedef unsigned short bf16_t;  // raw bf16 bit pattern in memory

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define VITA_WAVE 64

__device__ __forceinline__ float bf16_to_f32(bf16_t x) {
  return __uint_as_float(((unsigned)x) << 16);
}
// round-to-nearest-even, NaN preserved (hardware v_cvt_pk_bf16_f32)
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  __bf16 b = (__bf16)f;
  return *reinterpret_cast<bf16_t*>(&b);
}
// round a float to the nearest bf16 value, keep it as float (the "arrow" of the reference's
// unfused bf16 op chains)
__device__ __forceinline__ float bf16_round(float f) { return bf16_to_f32(f32_to_bf16(f)); }

__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  bf16x2 v;
  v[0] = (__bf16)lo;
  v[1] = (__bf16)hi;
  return *reinterpret_cast<unsigned*>(&v);
}
__device__ __forceinline__ float bf16lo_to_f32(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16hi_to_f32(unsigned w) { return __uint_as_float(w & 0xffff0000u); }

__device__ __forceinline__ float wave_reduce_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// block-wide sum for blockDim.x <= 1024 (<= 16 waves); `red` is >= 16 floats of LDS.
__device__ __forceinline__ float block_reduce_sum(float v, float* red) {
  v = wave_reduce_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();  // protect `red` from a previous use
  if (lane == 0) red[wid] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}

// RoPE (non-interleaved halves), bf16 rounding chain of rotary_pos_embedding.py:200-203 — shared by rope.hip and decode.hip
// rotate one pair of 8-wide vectors (x1 = first half, x2 = second half) with cos/sin vectors.
// ---- store of one 32 x 32 accumulator block of a "row per lane" epilogue (attention O, dQ, dK, dV) --------------------------------------
// The lane holds columns 8 rg + 4 hi .. + 3 (rg = 0 .. 3, hi = lane / 32) of ITS row in acc[4 rg .. 4 rg + 3]: lane i and lane i + 32 own
// the two 8-byte halves of every 16-byte column group.  r06: column groups (rg, rg + 1) are exchanged between the half-waves with
// v_permlane32_swap (lanes 32-63 of the first operand swap with lanes 0-31 of the second), after which the lower lanes hold group rg
// whole and the upper lanes group rg + 1 whole: two 16-byte stores per block instead of four 8-byte ones.  A workgroup's store tail is
// bound by store ISSUE, not bytes (MI355X_MICROARCH.md: ~9.3 k -> ~5.3 k cycles for the attention forward's shape).  `dst` = the row's
// first column of this block; it must be 16-byte aligned (the launchers check the strides).  Both lanes of a pair must be active.
// VITA_WIDE_STORE 0 builds the r05 form (same-box A / B).
#ifndef VITA_WIDE_STORE
#define VITA_WIDE_STORE 1
#endif
template <class Acc>
__device__ __forceinline__ void store_row_block32(bf16_t* dst, const Acc& acc, float scale, int hi) {
#if VITA_WIDE_STORE
#pragma unroll
  for (int pr = 0; pr < 2; ++pr) {
    const int k0 = 8 * pr, k1 = 8 * pr + 4;                      // accumulator registers of groups rg = 2 pr and 2 pr + 1
    const unsigned a0 = pack_bf16x2(acc[k0 + 0] * scale, acc[k0 + 1] * scale), a1 = pack_bf16x2(acc[k0 + 2] * scale, acc[k0 + 3] * scale);
    const unsigned b0 = pack_bf16x2(acc[k1 + 0] * scale, acc[k1 + 1] * scale), b1 = pack_bf16x2(acc[k1 + 2] * scale, acc[k1 + 3] * scale);
    const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
    const auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
    const u32x4 w = {r0[0], r1[0], r0[1], r1[1]};
    *reinterpret_cast<u32x4*>(dst + 16 * pr + 8 * hi) = w;
  }
#else
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) {
    const u32x2 w = {pack_bf16x2(acc[rg * 4 + 0] * scale, acc[rg * 4 + 1] * scale), pack_bf16x2(acc[rg * 4 + 2] * scale, acc[rg * 4 + 3] * scale)};
    *reinterpret_cast<u32x2*>(dst + 8 * rg + 4 * hi) = w;
  }
#endif
}

__device__ __forceinline__ void rope_rotate8(u32x4& x1, u32x4& x2, const u32x4& c, const u32x4& s,
                                             float sign) {
  u32x4 o1, o2;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float a0 = bf16lo_to_f32(x1[j]), a1 = bf16hi_to_f32(x1[j]);
    const float b0 = bf16lo_to_f32(x2[j]), b1 = bf16hi_to_f32(x2[j]);
    const float c0 = bf16lo_to_f32(c[j]), c1 = bf16hi_to_f32(c[j]);
    const float s0 = sign * bf16lo_to_f32(s[j]), s1 = sign * bf16hi_to_f32(s[j]);
    // out1 = x1*cos + (-x2)*sin ; out2 = x2*cos + x1*sin
    const float r10 = bf16_round(a0 * c0) + bf16_round(-b0 * s0);
    const float r11 = bf16_round(a1 * c1) + bf16_round(-b1 * s1);
    const float r20 = bf16_round(b0 * c0) + bf16_round(a0 * s0);
    const float r21 = bf16_round(b1 * c1) + bf16_round(a1 * s1);
    o1[j] = pack_bf16x2(r10, r11);
    o2[j] = pack_bf16x2(r20, r21);
  }
  x1 = o1;
  x2 = o2;
}

// Developer switches (VITA_ATTN_*, VITA_GEMM_*: kernel selection and timing aids) are honoured only when VITA_DEBUG
// is set in the environment; that flag is read ONCE per process, so a production process never consults them.
#include <stdlib.h>
static inline const char* vita_dev_getenv(const char* name) {
  static const bool on = [] { const char* d = getenv("VITA_DEBUG"); return d && d[0] && d[0] != '0'; }();
  return on ? getenv(name) : nullptr;
}

static inline int vita_check_launch() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? VITA_OK : VITA_ERR_LAUNCH;
}

// ---- LDS-DMA issued from inline asm ------------------------------------------------------------------------------------------
// `buffer_load_dword[x4] ... lds` moves 4 / 16 bytes per lane from a buffer address (descriptor base + per-lane byte offset) to
// LDS address m0 + lane * size.  Issued through the compiler's builtin, hipcc remembers that an LDS write is in flight and puts
// `s_waitcnt vmcnt(0)` in front of the next LDS read it cannot tell apart from it (every ds_read_b64_tr_b16, many ds_read_b128) —
// which turns a prefetch into a blocking load in the middle of the tile being computed.  From inline asm the compiler knows
// nothing about it; the kernels order DMA and reads themselves (counted `s_waitcnt vmcnt(n)` + barrier at stage boundaries).
typedef __attribute__((ext_vector_type(4))) unsigned int vita_rsrc_t;
__device__ __forceinline__ vita_rsrc_t vita_make_rsrc(const void* base) {      // raw buffer, no bounds, base must be wave-uniform
  const unsigned long long v = (unsigned long long)(uintptr_t)base;
  vita_rsrc_t r;
  r[0] = __builtin_amdgcn_readfirstlane((unsigned)v);
  r[1] = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32)) & 0xffffu;
  r[2] = 0x7fffffffu;
  r[3] = 0x00020000u;
  return r;
}
__device__ __forceinline__ vita_rsrc_t vita_make_rsrc_uniform(const void* base) {   // base already lives in SGPRs (no readfirstlane)
  const unsigned long long v = (unsigned long long)(uintptr_t)base;
  vita_rsrc_t r;
  r[0] = (unsigned)v; r[1] = (unsigned)(v >> 32) & 0xffffu; r[2] = 0x7fffffffu; r[3] = 0x00020000u;
  return r;
}
// m0 is handed to the asm as an INPUT operand ("{m0}"): the compiler writes it (and knows it did — no reserved-register clobber, so no
// "may not be preserved" warning and no reliance on m0 being dead across the statement); the `s_nop 0` is the wait state the hardware
// wants between an SALU write of m0 and an LDS-DMA that reads it, which the hazard recognizer cannot see inside an asm string.
__device__ __forceinline__ void vita_lds_dma16(vita_rsrc_t rsrc, unsigned voff_bytes, unsigned lds_addr) {
  asm volatile("s_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
               :: "{m0}"(lds_addr), "v"(voff_bytes), "s"(rsrc) : "memory");
}
__device__ __forceinline__ void vita_lds_dma4(vita_rsrc_t rsrc, unsigned voff_bytes, unsigned lds_addr) {
  asm volatile("s_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds"
               :: "{m0}"(lds_addr), "v"(voff_bytes), "s"(rsrc) : "memory");
}
