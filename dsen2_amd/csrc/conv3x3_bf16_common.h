// conv3x3_bf16_common.h — helpers shared by the bf16 body kernels (conv3x3_body16w.hip, conv3x3_body16x.hip):
// the inline-asm LDS-DMA statement, counted vmcnt waits and the 16 + 16 bit split of the fp32 residual stream.
#pragma once
#include "dsen2_internal.h"

namespace dsen2 {
namespace bf16k {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit field");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// One LDS-DMA wave instruction: 1 KiB global -> LDS, LDS address = m0v + 16 * lane, a lane whose offset is out of the
// descriptor's range writes zeros.  hipcc reserves M0 (it rejects "m0" in a clobber list as undefined behaviour), so
// the statement saves and restores it: the compiler's own uses of M0 never see the DMA's value.
__device__ __forceinline__ void lds_dma(unsigned m0v, unsigned voff, __amdgpu_buffer_rsrc_t rsrc, unsigned soff) {
  unsigned saved_m0;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(saved_m0) : "s"(m0v), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

// One dword per lane (256 B per wave instruction, LDS address = m0v + 4 * lane): small tables such as a layer's bias.
__device__ __forceinline__ void lds_dma_dword(unsigned m0v, unsigned voff, __amdgpu_buffer_rsrc_t rsrc, unsigned soff) {
  unsigned saved_m0;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dword %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(saved_m0) : "s"(m0v), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

// The same for lanes 0-15 only (the 16-slot tail of a 336-slot row): EXEC is narrowed and restored INSIDE the
// statement, so the compiler sees no divergent branch (a branch would cut the nine-step body into basic blocks).
__device__ __forceinline__ void lds_dma_low16(unsigned m0v, unsigned voff, __amdgpu_buffer_rsrc_t rsrc, unsigned soff) {
  unsigned saved_m0;
  unsigned long long saved_exec;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b64 %1, exec\n\ts_mov_b64 exec, 0xffff\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
               "buffer_load_dwordx4 %3, %4, %5 offen lds\n\ts_mov_b32 m0, %0\n\ts_mov_b64 exec, %1"
               : "=&s"(saved_m0), "=&s"(saved_exec) : "s"(m0v), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

__device__ __forceinline__ unsigned lds_address(const void* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p;
}

// ---- the 16 + 16 bit split of an fp32 value (see the header): two values per call ----
__device__ __forceinline__ void split2(unsigned u0, unsigned u1, unsigned& hi, unsigned& lo) {
  lo = __builtin_amdgcn_perm(u1, u0, 0x05040100u);                    // [u1.lo16 : u0.lo16]
  const unsigned top = __builtin_amdgcn_perm(u1, u0, 0x07060302u);    // [u1.hi16 : u0.hi16]
  const u16x2 r = __builtin_bit_cast(u16x2, top) + (__builtin_bit_cast(u16x2, lo) >> (unsigned short)15);
  hi = __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ void join2(unsigned hi, unsigned lo, unsigned& u0, unsigned& u1) {
  const u16x2 t = __builtin_bit_cast(u16x2, hi) - (__builtin_bit_cast(u16x2, lo) >> (unsigned short)15);
  const unsigned top = __builtin_bit_cast(unsigned, t);
  u0 = __builtin_amdgcn_perm(top, lo, 0x05040100u);
  u1 = __builtin_amdgcn_perm(top, lo, 0x07060302u);
}

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

}  // namespace bf16k
}  // namespace dsen2
