// TEST INFRASTRUCTURE.  A stand-in for <hip/hip_runtime.h> that lets g++ compile moshpp_amd/csrc/{moshii_api,chain_solve,lbs_forward,stagei}.hip UNCHANGED
// and run their kernels on the CPU: every workgroup is executed as one fiber per thread (ucontext), with real barrier semantics for
// __syncthreads / s_barrier and wave-level rendezvous for the cross-lane operations the kernels use (__shfl_down, readlane,
// wave_barrier).  "Device" memory is host memory.  Used only by tests/emu (build_chain_emu.py); the product is built by hipcc.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

struct dim3 { unsigned x, y, z; dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };
struct int2 { int x, y; };
struct alignas(16) int4 { int x, y, z, w; };
struct alignas(16) double2 { double x, y; };
inline double2 make_double2(double x, double y) { double2 r; r.x = x; r.y = y; return r; }
struct alignas(16) double4 { double x, y, z, w; };

namespace hipemu {
struct ThreadCtx { dim3 tid, bid, bdim, gdim; };
extern thread_local ThreadCtx* cur_ptr;             // the running fiber's indices (set by the scheduler at every switch)
inline ThreadCtx& cur() { return *cur_ptr; }
void barrier();                                     // workgroup barrier
void wave_sync();                                   // all lanes of the caller's wavefront rendezvous
double wave_exchange(double v, int src_lane);       // value of `v` in lane `src_lane` of the caller's wavefront (all lanes call)
void wave_gather2(double a, double b, const double** all);   // every lane's (a, b) of the caller's wavefront: all[0][2*lane], all[0][2*lane+1]
const char* wave_allgather(const void* mine, int nbytes);    // every lane's `nbytes` (<= 64) of the caller's wavefront, lane-major, 64 bytes apart
extern const char* launch_name;                     // (HIPEMU_PROF=1: per-kernel wall time at exit)
void launch(dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()>& body);
void register_dynamic_lds(double* base, size_t bytes);   // arrays behind `extern __shared__`: guarded beyond the launch's lds_bytes
}

#define MOSHII_EMULATION 1                           // (moshii_api.hip: cooperative chains only on request -- see HIPEMU_CONCURRENT)
#define __global__
#define __device__
#define __host__
#define __shared__ thread_local                     // one OS thread runs one workgroup at a time (HIPEMU_CONCURRENT=1: one OS thread PER workgroup of a launch): thread_local == per-workgroup
#define __constant__ static
#define __forceinline__ inline
#define __noinline__ __attribute__((noinline))
#define __launch_bounds__(...)
#define MOSHII_AS_GLOBAL                             // (address spaces: nothing to say on the CPU)
#ifndef HIPEMU_NATIVE_F16                            // lbs_forward.hip is compiled by clang++ (ext_vector_type, _Float16 arithmetic)
#define _Float16 unsigned short                      // g++ 11 translation units: only pointer members of Lbs32Model
#endif
#define address_space(n)                             // __attribute__((address_space(1))) -> __attribute__(())
#define HIP_SYMBOL(x) (&(x))
#define threadIdx (hipemu::cur().tid)
#define blockIdx (hipemu::cur().bid)
#define blockDim (hipemu::cur().bdim)
#define gridDim (hipemu::cur().gdim)

typedef int hipError_t;
typedef void* hipStream_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorUnknown = 999 };
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipDeviceAttributeMultiprocessorCount = 1, hipFuncAttributeMaxDynamicSharedMemorySize = 2 };

template <class T> inline hipError_t hipMalloc(T** p, size_t n) { *p = (T*)calloc(1, n ? n : 1); return *p ? hipSuccess : hipErrorUnknown; }
inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (n) memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { if (n) memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpy2DAsync(void* d, size_t dpitch, const void* s, size_t spitch, size_t width, size_t height, hipMemcpyKind, hipStream_t) {
    for (size_t r = 0; r < height; ++r) memcpy((char*)d + r * dpitch, (const char*)s + r * spitch, width);
    return hipSuccess;
}
inline hipError_t hipMemset(void* d, int v, size_t n) { if (n) memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { if (n) memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemcpyFromSymbol(void* d, const void* sym, size_t n) { memcpy(d, sym, n); return hipSuccess; }
inline hipError_t hipMemcpyToSymbol(void* sym, const void* s, size_t n) { memcpy(sym, s, n); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t) { return "emulated"; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
inline hipError_t hipDeviceGetAttribute(int* v, int, int) { *v = 8; return hipSuccess; }     // 8 "CUs": small chunk counts in the emulated runs
inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }
#define hipLaunchKernelGGL(kern, grid, block, lds, stream, ...) \
    (hipemu::launch_name = #kern, hipemu::launch(grid, block, lds, [&]() { kern(__VA_ARGS__); }))

inline void __syncthreads() { hipemu::barrier(); }
template <class T> inline T __shfl_down(T v, int delta, int width = 64) {
    int lane = (int)(hipemu::cur().tid.x % 64);
    int src = lane + delta;
    double got = hipemu::wave_exchange((double)v, (src < width && src < 64) ? src : lane);
    return (T)got;
}
template <class T> inline T __shfl(T v, int src, int width = 64) {
    const int lane = (int)(hipemu::cur().tid.x % 64);
    return (T)hipemu::wave_exchange((double)v, (lane & ~(width - 1)) | (src & (width - 1)));
}
inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }   // (HIPEMU_CONCURRENT=1: workgroups of a launch run side by side)
inline int __double2loint(double v) { int64_t b; memcpy(&b, &v, 8); return (int)(b & 0xffffffff); }
inline int __double2hiint(double v) { int64_t b; memcpy(&b, &v, 8); return (int)(b >> 32); }
inline double __hiloint2double(int hi, int lo) { int64_t b = ((int64_t)hi << 32) | (uint32_t)lo; double v; memcpy(&v, &b, 8); return v; }
inline int __builtin_amdgcn_readlane(int v, int lane) {          // exact for 32-bit payloads: a double carries any int exactly
    return (int)hipemu::wave_exchange((double)v, lane);
}
// DPP moves: the lane controls the kernels use (quad_perm, row_mirror, row_half_mirror), all rows and banks enabled
inline int __builtin_amdgcn_update_dpp(int, int src, int ctrl, int, int, bool) {
    const int lane = (int)(hipemu::cur().tid.x % 64);
    int from;
    if (ctrl < 0x100) from = (lane & ~3) | ((ctrl >> (2 * (lane & 3))) & 3);
    else if (ctrl == 0x140) from = (lane & ~15) | (15 - (lane & 15));
    else if (ctrl == 0x141) from = (lane & ~7) | (7 - (lane & 7));
    else { fprintf(stderr, "hipemu: DPP control 0x%x is not emulated\n", ctrl); abort(); }
    return (int)hipemu::wave_exchange((double)src, from);
}
inline unsigned long long __ballot(int pred) {
    unsigned long long m = 0;
    for (int l = 0; l < 64; ++l) if (hipemu::wave_exchange(pred ? 1.0 : 0.0, l) != 0.0) m |= 1ull << l;
    return m;
}
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __builtin_amdgcn_readfirstlane(int v) { return v; }   // only used to scalarise wave-uniform values
// v_mfma_f64_16x16x4_f64: D = A (16x4) . B (4x16) + C; lane l supplies A[l & 15][l >> 4] and B[l >> 4][l & 15], register i of
// lane l holds C/D[(l >> 4) + 4 i][l & 15]; the sum over k is taken as an in-order fma chain
typedef double hipemu_v4d __attribute__((vector_size(32)));
inline hipemu_v4d __builtin_amdgcn_mfma_f64_16x16x4f64(double a, double b, hipemu_v4d c, int, int, int) {
    const double* all = nullptr;
    hipemu::wave_gather2(a, b, &all);
    const int lane = (int)(hipemu::cur().tid.x % 64), n = lane & 15;
    hipemu_v4d d = c;
    for (int i = 0; i < 4; ++i) {
        const int m = (lane >> 4) + 4 * i;
        double acc = c[i];
        for (int k = 0; k < 4; ++k) acc = std::fma(all[2 * (m + 16 * k)], all[2 * (n + 16 * k) + 1], acc);
        d[i] = acc;
    }
    return d;
}
#ifdef HIPEMU_NATIVE_F16
// v_mfma_f32_16x16x4_f32: D = A (16x4) . B (4x16) + C; lane l supplies A[l & 15][l >> 4] and B[l >> 4][l & 15], register r of lane l
// holds C/D[4 (l >> 4) + r][l & 15]; the sum over k as an in-order f32 fma chain
typedef float hipemu_f4s __attribute__((ext_vector_type(4)));
inline hipemu_f4s __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, hipemu_f4s c, int, int, int) {
    const double* all = nullptr;
    hipemu::wave_gather2((double)a, (double)b, &all);
    const int lane = (int)(hipemu::cur().tid.x % 64), n = lane & 15;
    hipemu_f4s d = c;
    for (int r = 0; r < 4; ++r) {
        const int m = 4 * (lane >> 4) + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) acc = std::fma((float)all[2 * (m + 16 * k)], (float)all[2 * (n + 16 * k) + 1], acc);
        d[r] = acc;
    }
    return d;
}
// v_mfma_f32_16x16x32_f16: D = A (16x32) . B (32x16) + C; lane l supplies A[l & 15][8 (l >> 4) + e] and B[8 (l >> 4) + e][l & 15],
// register r of lane l holds C/D[4 (l >> 4) + r][l & 15]; products are exact in f32, the sum is taken in double and rounded once
typedef _Float16 hipemu_h8 __attribute__((ext_vector_type(8)));
typedef float hipemu_f4 __attribute__((ext_vector_type(4)));
inline hipemu_f4 __builtin_amdgcn_mfma_f32_16x16x32_f16(hipemu_h8 a, hipemu_h8 b, hipemu_f4 c, int, int, int) {
    struct { hipemu_h8 a, b; } mine = {a, b};
    const char* all = hipemu::wave_allgather(&mine, 32);
    const int lane = (int)(hipemu::cur().tid.x % 64), n = lane & 15;
    hipemu_f4 d = c;
    for (int r = 0; r < 4; ++r) {
        const int m = 4 * (lane >> 4) + r;
        double acc = c[r];
        for (int kq = 0; kq < 4; ++kq) {
            const _Float16* pa = reinterpret_cast<const _Float16*>(all + (size_t)(m + 16 * kq) * 64);
            const _Float16* pb = reinterpret_cast<const _Float16*>(all + (size_t)(n + 16 * kq) * 64 + 16);
            for (int e = 0; e < 8; ++e) acc += (double)((float)pa[e] * (float)pb[e]);
        }
        d[r] = (float)acc;
    }
    return d;
}
// LDS-DMA: every lane's `size` bytes land at the wave-uniform LDS address + lane * size (immediately here; the kernels' waits and
// barriers are what the device build relies on)
inline void __builtin_amdgcn_global_load_lds(const void* g, void* l, int size, int offset, int) {
    const int lane = (int)(hipemu::cur().tid.x % 64);
    memcpy((char*)l + offset + (size_t)lane * size, (const char*)g + offset, size);
}
#endif
// buffer loads / stores: resource = base pointer; a lane's 16 bytes at base + voffset + soffset
#ifdef HIPEMU_NATIVE_F16
struct __amdgpu_buffer_rsrc_t { char* base; };
typedef unsigned hipemu_u4 __attribute__((ext_vector_type(4)));
inline __amdgpu_buffer_rsrc_t __builtin_amdgcn_make_buffer_rsrc(void* p, short, int, int) { return __amdgpu_buffer_rsrc_t{(char*)p}; }
inline hipemu_u4 __builtin_amdgcn_raw_buffer_load_b128(__amdgpu_buffer_rsrc_t rs, unsigned voffset, unsigned soffset, int) {
    hipemu_u4 v; memcpy(&v, rs.base + voffset + soffset, 16); return v;
}
inline unsigned __builtin_amdgcn_raw_buffer_load_b32(__amdgpu_buffer_rsrc_t rs, unsigned voffset, unsigned soffset, int) {
    unsigned v; memcpy(&v, rs.base + voffset + soffset, 4); return v;
}
inline void __builtin_amdgcn_raw_buffer_store_b128(hipemu_u4 v, __amdgpu_buffer_rsrc_t rs, unsigned voffset, unsigned soffset, int) {
    memcpy(rs.base + voffset + soffset, &v, 16);
}
#endif
inline void __builtin_amdgcn_s_waitcnt(int) {}
inline void __builtin_amdgcn_s_nop(int) {}
inline unsigned __builtin_amdgcn_s_getreg(int) { return 0; }   // (hardware ids: everything runs on "CU 0")
inline void __builtin_amdgcn_sched_barrier(int) {}
inline float __int_as_float(int v) { float f; memcpy(&f, &v, 4); return f; }
inline double __builtin_amdgcn_rcp(double x) { return 1.0 / x; }
inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
inline void __builtin_amdgcn_wave_barrier() { hipemu::wave_sync(); }
inline void __builtin_amdgcn_fence(int, const char*) {}
void hipemu_yield();                                  // (a spinning workgroup lets the other workgroups' OS threads run)
inline void __builtin_amdgcn_s_sleep(int) { hipemu_yield(); }
// device-scope atomics / fences of the inter-workgroup protocols: real (sequentially consistent) atomics, so that workgroups running
// side by side (HIPEMU_CONCURRENT=1: the cooperative chains) see each other's flags; plain semantics otherwise
#define __HIP_MEMORY_SCOPE_AGENT 0
#define __hip_atomic_load(p, order, scope) __atomic_load_n((p), __ATOMIC_SEQ_CST)
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n((p), (v), __ATOMIC_SEQ_CST)
inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
inline long long clock64() { return 0; }
inline long long wall_clock64() { return 0; }
inline double rsqrt(double x) { return 1.0 / sqrt(x); }
inline int min(int a, int b) { return a < b ? a : b; }
inline int max(int a, int b) { return a > b ? a : b; }
inline double min(double a, double b) { return a < b ? a : b; }
inline double max(double a, double b) { return a > b ? a : b; }
inline size_t min(size_t a, size_t b) { return a < b ? a : b; }
inline size_t max(size_t a, size_t b) { return a > b ? a : b; }

// dynamic LDS: the kernels declare `extern __shared__ double lds[]` (namespace moshii, chain_solve.hip) and `... sm[]` (unnamed
// namespace, moshii_api.hip); with __shared__ = thread_local these block-scope externs need a definition in their namespace
namespace moshii { alignas(16) inline thread_local double lds[160 * 1024 / 8]; }
namespace { alignas(16) thread_local double sm[160 * 1024 / 8]; }
#ifdef HIPEMU_UNNAMED_LDS   // stagei.hip: `extern __shared__ double lds[]` inside its unnamed namespace (k_s1_elim)
namespace { alignas(16) thread_local double lds[160 * 1024 / 8]; }
// (g++ routes the kernel's block-scope `extern thread_local` through the unit's TLS init function and only emits that function when
//  some thread_local of the unit has a dynamic initialiser: give it one)
namespace { inline int hipemu_tls_anchor_value() { return 1; } thread_local int hipemu_tls_anchor = hipemu_tls_anchor_value(); }
namespace { struct HipEmuLdsRegistration2 { HipEmuLdsRegistration2() { hipemu::register_dynamic_lds(lds, sizeof(lds)); } } hipemu_lds_registration2; }
#endif
#ifdef HIPEMU_NATIVE_F16   // lbs_forward.hip: `extern __shared__ char lds_raw[]` (export kernel), `... float smf[]` (plain kernel)
namespace { alignas(16) thread_local char lds_raw[160 * 1024]; alignas(16) thread_local float smf[160 * 1024 / 4]; }
namespace { struct HipEmuLdsRegistration { HipEmuLdsRegistration() { hipemu::register_dynamic_lds((double*)lds_raw, sizeof(lds_raw)); hipemu::register_dynamic_lds((double*)smf, sizeof(smf)); } } hipemu_lds_registration; }
#else
namespace { struct HipEmuLdsRegistration { HipEmuLdsRegistration() { hipemu::register_dynamic_lds(moshii::lds, sizeof(moshii::lds)); hipemu::register_dynamic_lds(sm, sizeof(sm)); } } hipemu_lds_registration; }
#endif

// the one inline-assembly statement of the chain kernel is a workgroup barrier ("s_waitcnt ...; s_barrier"):
//   asm volatile("..." ::: "memory")  ->  hipemu::barrier()
#define asm
#define volatile(...) hipemu::barrier()
