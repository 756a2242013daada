// gload.hpp — global loads the compiler does not see (inline asm), for hand-pipelined loops (count3.hip, apply3.hip).
//
// Why: in a software-pipelined loop the loads of the NEXT block must stay in flight while the current block is worked on.  The
// compiler, however, (a) loads into temporaries and copies them into the loop-carried registers - the copy needs the data, so it
// waits (s_waitcnt vmcnt(0)) right behind the load - and (b) cannot count loads that sit in conditional code, so every wait it
// inserts is vmcnt(0).  With the loads written as asm the destination registers are the loop-carried ones themselves, the address is
// the wave-uniform base + 32-bit lane offset form, and the ONE wait per loop trip is placed by hand (gwait), behind the block's work.
// The compiler's own loads (rare paths) still drain the counter when they are waited for - harmless, only less overlap.
#pragma once
#include <cstdint>

namespace elp {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// a wave-uniform value the compiler holds in vector registers (it came out of LDS, say) moved to scalar registers: the "s" operands below
__device__ __forceinline__ uint32_t to_sgpr(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
__device__ __forceinline__ uint64_t to_sgpr(uint64_t x) { return (uint64_t)to_sgpr((uint32_t)x) | ((uint64_t)to_sgpr((uint32_t)(x >> 32)) << 32); }
// base (wave-uniform, in scalar registers) + off (32-bit, per lane)
__device__ __forceinline__ void gload_x4(u32x4 &v, const void *sbase, uint32_t off) {
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(off), "s"(sbase) : "memory");
}
__device__ __forceinline__ void gload_x2(u32x2 &v, const void *sbase, uint32_t off) {
  asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(v) : "v"(off), "s"(sbase) : "memory");
}
__device__ __forceinline__ void gload_x1(uint32_t &v, const void *sbase, uint32_t off) {
  asm volatile("global_load_dword %0, %1, %2" : "=v"(v) : "v"(off), "s"(sbase) : "memory");
}
// per-lane 64-bit address
__device__ __forceinline__ void gload_x4(u32x4 &v, uint64_t addr) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(addr) : "memory"); }
__device__ __forceinline__ void gload_x1(uint32_t &v, uint64_t addr) { asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(addr) : "memory"); }
__device__ __forceinline__ void gload_x2(u32x2 &v, uint64_t addr) { asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v) : "v"(addr) : "memory"); }

// a store the compiler does not see either: with every VMEM instruction of a loop trip written by hand the trip's wait can be COUNTED
// (gwait_but<N>): stores share the vmcnt counter with loads on gfx9 and complete in issue order with them, so a vmcnt(0) behind a
// block's store would also wait for that store's acknowledgement - a full write round trip exposed in every trip
__device__ __forceinline__ void gstore_x4(uint64_t addr, u32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(addr), "v"(v) : "memory"); }

// The wait.  It names NO registers on purpose: an operand tied to a loaded register ("+v") lets the compiler copy that register into
// the operand's register IN FRONT of the statement - i.e. before the data has arrived (seen in the ISA of the first version: the
// record registers were copied two instructions ahead of the s_waitcnt, and runs differed from each other at 12 M reads).  The caller
// follows it with an EMPTY asm statement that takes the loaded registers as "+v" operands: volatile statements keep their order, so
// whatever copies the compiler wants sit behind the wait, and every use of the loaded values depends on that second statement.
// tests/test_asm_pipeline.py checks the generated ISA: nothing reads a load's destination between the load and the wait.
__device__ __forceinline__ void gwait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// waits until at most N vector-memory instructions are outstanding: the N youngest (which the caller knows to be stores issued by hand,
// unconditionally, behind every load it is waiting for) may still be on their way
template <int N>
__device__ __forceinline__ void gwait_but() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }
// A copy the compiler cannot move in front of the wait (volatile statements keep their order): for loaded registers whose values move
// on to other registers while the buffer they landed in is loaded again (the record buffers of count3 / apply3).
__device__ __forceinline__ uint32_t amov(uint32_t x) {
  uint32_t r;
  asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(x));
  return r;
}
__device__ __forceinline__ u32x4 amov(u32x4 x) { return (u32x4){amov(x.x), amov(x.y), amov(x.z), amov(x.w)}; }
__device__ __forceinline__ u32x2 amov(u32x2 x) { return (u32x2){amov(x.x), amov(x.y)}; }

}  // namespace elp
