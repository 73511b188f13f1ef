// TEST INFRASTRUCTURE ONLY (tests/cpp/kemu): a small CPU stand-in for the part of the HIP device programming model the kernels of
// gyeeta_amd/csrc use, so that a kernel's LOGIC can be run under g++ without a GPU and compared with the oracle.  It is found as
// <hip/hip_runtime.h> by putting tests/cpp/kemu first on the include path; nothing under gyeeta_amd/ ever includes it.
//
// Model: one OS thread per GPU thread, one workgroup at a time.  __shared__ = a function-local static (one copy, the workgroups run one
// after the other); __syncthreads() = a barrier of the workgroup's live threads; the wave64 intrinsics (__shfl*, __ballot,
// readfirstlane) exchange through a per-wave buffer with a barrier of the wave's live threads on either side, i.e. they must be
// reached by every live lane of the wave (true of the kernels tested: their wave operations sit in wave-uniform control flow).
// A thread that returns from the kernel leaves the barriers (as an exited wave leaves s_barrier).  Atomics are the host's.
// What it does NOT model: the memory model (everything is sequentially consistent here), execution masks inside a wave operation,
// LDS size limits, timing.  The GPU parity tests (-m gpu) remain the check of the real thing.
#pragma once

#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __constant__ static const
#define __launch_bounds__(...)
#define __align__(n) __attribute__((aligned(n)))

struct uint2 { uint32_t x, y; };
struct uint4 { uint32_t x, y, z, w; };
struct int2 { int32_t x, y; };
struct ulonglong2 { unsigned long long x, y; };
static inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
static inline int2 make_int2(int32_t x, int32_t y) { return int2{x, y}; }

namespace kemu {

struct Dim3 { uint32_t x = 1, y = 1, z = 1; };

struct DynBarrier { // a barrier whose participants may leave (C++20: the waiters sleep on the generation word, not on a condition variable)
	std::mutex m;
	int expected = 0, waiting = 0;
	std::atomic<uint32_t> gen{0};
	void reset(int n) { expected = n; waiting = 0; }
	void arrive_wait()
	{
		uint32_t g;
		bool release = false;
		{
			std::lock_guard<std::mutex> l(m);
			g = gen.load(std::memory_order_relaxed);
			if (++waiting >= expected) {
				waiting = 0;
				gen.store(g + 1, std::memory_order_release);
				release = true;
			}
		}
		if (release) {
			gen.notify_all();
		} else {
			while (gen.load(std::memory_order_acquire) == g) gen.wait(g, std::memory_order_acquire);
		}
	}
	void leave()
	{
		bool release = false;
		{
			std::lock_guard<std::mutex> l(m);
			--expected;
			if (expected > 0 && waiting >= expected) {
				waiting = 0;
				gen.fetch_add(1, std::memory_order_release);
				release = true;
			}
		}
		if (release) gen.notify_all();
	}
};

struct Wave {
	DynBarrier bar;
	uint64_t x[64];
};

struct Block {
	DynBarrier bar;
	std::vector<Wave> waves;
	std::vector<uint8_t> dyn_lds;
};

inline Block *g_block = nullptr;
inline Dim3 g_blockDim, g_gridDim;
inline thread_local Dim3 t_threadIdx, t_blockIdx;

inline Wave &wave() { return g_block->waves[t_threadIdx.x >> 6]; }
inline uint32_t lane() { return t_threadIdx.x & 63u; }
inline void *dyn_lds() { return g_block->dyn_lds.data(); }

// can this process have `n` threads at once (container limits differ)?  The tests exit with status 77 (reported as a skip) if not.
inline bool can_run(uint32_t n)
{
	std::vector<std::thread> th;
	std::atomic<bool> go{false};
	bool ok = true;
	try {
		th.reserve(n);
		for (uint32_t t = 0; t < n; ++t)
			th.emplace_back([&] {
				while (!go.load(std::memory_order_acquire)) std::this_thread::yield();
			});
	} catch (...) {
		ok = false;
	}
	go.store(true, std::memory_order_release);
	for (auto &x : th) x.join();
	return ok;
}

// run `kernel()` for grid x block threads (1-D), one workgroup after the other
template <class F>
void launch(uint32_t grid, uint32_t block, size_t dyn_lds_bytes, F kernel)
{
	g_blockDim.x = block;
	g_gridDim.x = grid;
	for (uint32_t b = 0; b < grid; ++b) {
		Block blk;
		const uint32_t nw = (block + 63u) / 64u;
		blk.waves = std::vector<Wave>(nw);
		blk.bar.reset((int)block);
		for (uint32_t w = 0; w < nw; ++w) blk.waves[w].bar.reset((int)std::min<uint32_t>(64u, block - 64u * w));
		blk.dyn_lds.assign(dyn_lds_bytes + 16, 0);
		g_block = &blk;
		std::vector<std::thread> th;
		th.reserve(block);
		for (uint32_t t = 0; t < block; ++t)
			th.emplace_back([&, t, b] {
				t_threadIdx.x = t;
				t_blockIdx.x = b;
				kernel();
				blk.waves[t >> 6].bar.leave();
				blk.bar.leave();
			});
		for (auto &x : th) x.join();
		g_block = nullptr;
	}
}

template <class T>
inline uint64_t to_raw(T v)
{
	static_assert(sizeof(T) <= 8, "wave exchange of up to 8 bytes");
	uint64_t r = 0;
	memcpy(&r, &v, sizeof(T));
	return r;
}
template <class T>
inline T from_raw(uint64_t r)
{
	T v;
	memcpy(&v, &r, sizeof(T));
	return v;
}
template <class T>
inline T exchange(T v, uint32_t src)
{
	Wave &w = wave();
	w.x[lane()] = to_raw(v);
	w.bar.arrive_wait();
	const uint64_t r = w.x[src & 63u];
	w.bar.arrive_wait();
	return from_raw<T>(r);
}

} // namespace kemu

#define threadIdx kemu::t_threadIdx
#define blockIdx kemu::t_blockIdx
#define blockDim kemu::g_blockDim
#define gridDim kemu::g_gridDim

static inline void __syncthreads() { kemu::g_block->bar.arrive_wait(); }

template <class T>
static inline T __shfl_xor(T v, int d, int width = 64)
{
	const uint32_t l = kemu::lane();
	uint32_t src = l ^ (uint32_t)d;
	if (src / (uint32_t)width != l / (uint32_t)width) src = l;
	return kemu::exchange(v, src);
}
template <class T>
static inline T __shfl_up(T v, unsigned d, int width = 64)
{
	const uint32_t l = kemu::lane();
	const uint32_t base = l / (uint32_t)width * (uint32_t)width;
	const uint32_t src = (l >= base + d) ? l - d : l;
	return kemu::exchange(v, src);
}
template <class T>
static inline T __shfl(T v, int srcLane, int width = 64)
{
	const uint32_t l = kemu::lane();
	const uint32_t base = l / (uint32_t)width * (uint32_t)width;
	return kemu::exchange(v, base + ((uint32_t)srcLane % (uint32_t)width));
}
static inline unsigned long long __ballot(int pred)
{
	kemu::Wave &w = kemu::wave();
	w.x[kemu::lane()] = pred ? 1u : 0u;
	w.bar.arrive_wait();
	unsigned long long m = 0;
	const int live = 64; // (lanes that left the kernel keep their last word: kernels tested do not ballot after partial exits)
	for (int i = 0; i < live; ++i)
		if (w.x[i]) m |= 1ull << i;
	w.bar.arrive_wait();
	return m;
}
static inline int __builtin_amdgcn_readfirstlane(int v) { return __shfl(v, 0, 64); }
#define __builtin_amdgcn_fence(...) ((void)0)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
static inline void __builtin_amdgcn_wave_barrier() { kemu::wave().bar.arrive_wait(); }

// fast-math device intrinsics (only the synthetic event generator uses them)
#define __logf(x) logf(x)
#define __cosf(x) cosf(x)
#define __sinf(x) sinf(x)
#define __expf(x) expf(x)

static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __popc(unsigned int x) { return __builtin_popcount(x); }
static inline int __clz(int x) { return x ? __builtin_clz((unsigned)x) : 32; }
static inline int __clzll(long long x) { return x ? __builtin_clzll((unsigned long long)x) : 64; }
static inline int __ffsll(long long x) { return __builtin_ffsll(x); }
static inline unsigned long long __umul64hi(unsigned long long a, unsigned long long b) { return (unsigned long long)(((unsigned __int128)a * b) >> 64); }

template <class T, class U>
static inline T atomicAdd(T *p, U v) { return __atomic_fetch_add(p, (T)v, __ATOMIC_SEQ_CST); }
template <class T, class U>
static inline T atomicOr(T *p, U v) { return __atomic_fetch_or(p, (T)v, __ATOMIC_SEQ_CST); }
template <class T, class U>
static inline T atomicMax(T *p, U v)
{
	T old = __atomic_load_n(p, __ATOMIC_SEQ_CST);
	while (old < (T)v && !__atomic_compare_exchange_n(p, &old, (T)v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {
	}
	return old;
}
template <class T, class U>
static inline T atomicMin(T *p, U v)
{
	T old = __atomic_load_n(p, __ATOMIC_SEQ_CST);
	while (old > (T)v && !__atomic_compare_exchange_n(p, &old, (T)v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {
	}
	return old;
}
template <class T, class U, class V>
static inline T atomicCAS(T *p, U cmp, V val)
{
	T expected = (T)cmp;
	__atomic_compare_exchange_n(p, &expected, (T)val, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);
	return expected;
}

template <class T>
static inline T min(T a, T b) { return b < a ? b : a; }
template <class T>
static inline T max(T a, T b) { return a < b ? b : a; }
