// Stand-in for the reference's src/pch.h (which includes <Windows.h>, <wrl.h>): the same typedefs and helper
// templates, nothing of the Win32 side.  Test infrastructure (oracle/refbuild) — prefix header of every reference unit.
#pragma once
#include <cstdint>
#include <cstring>
#include <cmath>
#include <math.h>
#include <cfloat>
#include <climits>
#include <cassert>
#include <limits>
#include <array>
#include <string>
#include <vector>
#include <iostream>
#include <memory>
#include <mutex>
#include <filesystem>
#include <functional>
#include <unordered_map>
#include <algorithm>
#include <immintrin.h>

typedef int8_t int8; typedef uint8_t uint8; typedef int16_t int16; typedef uint16_t uint16;
typedef int32_t int32; typedef uint32_t uint32; typedef int64_t int64; typedef uint64_t uint64; typedef wchar_t wchar;

#define ASSERT(cond) ((void)0)
template <typename T> using ref = std::shared_ptr<T>;
template <typename T> using weakref = std::weak_ptr<T>;
template <typename T, typename... Args> inline ref<T> make_ref(Args&&... args) { return std::make_shared<T>(std::forward<Args>(args)...); }
#define arraysize(arr) (sizeof(arr) / sizeof((arr)[0]))
template <typename T> constexpr inline auto min(T a, T b) { return (a < b) ? a : b; }
template <typename T> constexpr inline auto max(T a, T b) { return (a < b) ? b : a; }
template <auto V> static constexpr auto force_consteval = V;
#define setBit(mask, bit) (mask) |= (1 << (bit))
#define unsetBit(mask, bit) (mask) ^= (1 << (bit))

// MSVC's __m128::m128_f32 / __m128i::m128i_i32 members
template <typename T> static inline float* REF_LANEF(const T& v) { return (float*)&v; }
template <typename T> static inline int* REF_LANEI(const T& v) { return (int*)&v; }
// SVML integer division (not called by the routines under test)
static inline __m128i ref_div_epi32_128(__m128i a, __m128i b) { alignas(16) int x[4], y[4]; _mm_store_si128((__m128i*)x, a); _mm_store_si128((__m128i*)y, b); for (int i = 0; i < 4; ++i) x[i] /= y[i]; return _mm_load_si128((__m128i*)x); }
static inline __m256i ref_div_epi32_256(__m256i a, __m256i b) { alignas(32) int x[8], y[8]; _mm256_store_si256((__m256i*)x, a); _mm256_store_si256((__m256i*)y, b); for (int i = 0; i < 8; ++i) x[i] /= y[i]; return _mm256_load_si256((__m256i*)x); }
// MSVC CRT / intrinsics spellings
#define __debugbreak() ((void)0)
#define _BitScanForward(idx, mask) ((mask) ? (*(idx) = (unsigned long)__builtin_ctz(mask), 1) : 0)
#define _BitScanReverse(idx, mask) ((mask) ? (*(idx) = (unsigned long)(31 - __builtin_clz(mask)), 1) : 0)

// ---- Win32 virtual-memory calls used by the reference's memory_arena (src/core/memory.cpp) ----
#include <sys/mman.h>
#include <unistd.h>
#define MEM_RESERVE 0x2000
#define MEM_COMMIT 0x1000
#define MEM_RELEASE 0x8000
#define PAGE_READWRITE 0x04
struct SYSTEM_INFO { unsigned long dwPageSize; };
static inline void GetSystemInfo(SYSTEM_INFO* s) { s->dwPageSize = (unsigned long)sysconf(_SC_PAGESIZE); }
namespace ref_vm { struct region { void* base; size_t size; }; inline std::vector<region>& regions() { static std::vector<region> r; return r; } }
static inline void* VirtualAlloc(void* address, size_t size, unsigned type, unsigned /*protect*/)
{
	if (type == MEM_RESERVE)
	{
		void* p = mmap(nullptr, size, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
		if (p == MAP_FAILED) { return nullptr; }
		ref_vm::regions().push_back({ p, size });
		return p;
	}
	// MEM_COMMIT: fresh pages, zero-filled like Windows commits them
	mprotect(address, size, PROT_READ | PROT_WRITE);
	return address;
}
static inline int VirtualFree(void* address, size_t, unsigned)
{
	auto& r = ref_vm::regions();
	for (size_t i = 0; i < r.size(); ++i) { if (r[i].base == address) { munmap(address, r[i].size); r.erase(r.begin() + i); return 1; } }
	return 0;
}
namespace fs = std::filesystem;

// ---- transcendental functions: the reference's float calls of acos / atan2 / sin / cos go to the oracle's fixed operation
// sequences (oracle/ora_det.cpp, linked into libref.so) instead of glibc, because the last ulp of libm differs between C
// runtimes and would otherwise be the only difference between the reference and the restatement.  Double and SIMD
// overloads keep their own definitions (the function-like macros rename the whole overload set consistently).
namespace ora { float det_atan2f(float y, float x); float det_acosf(float x); float det_sinf(float x); float det_cosf(float x); }
static inline float ref_acos(float x) { return ora::det_acosf(x); }
static inline double ref_acos(double x) { return ::acos(x); }
static inline float ref_atan2(float y, float x) { return ora::det_atan2f(y, x); }
static inline double ref_atan2(double y, double x) { return ::atan2(y, x); }
static inline float ref_sin(float x) { return ora::det_sinf(x); }
static inline double ref_sin(double x) { return ::sin(x); }
static inline float ref_cos(float x) { return ora::det_cosf(x); }
static inline double ref_cos(double x) { return ::cos(x); }
#ifndef REF_NATIVE_LIBM
#define acos(x) ref_acos(x)
#define atan2(y, x) ref_atan2(y, x)
#define sin(x) ref_sin(x)
#define cos(x) ref_cos(x)
#endif
#define __declspec(x)
