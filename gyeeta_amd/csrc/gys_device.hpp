// gys_device.hpp -- device-side primitives of libgysketch (gfx950 / CDNA4, wave64).
//
// Everything here is integer/byte work on the reference's own definitions; citations are file:line under the Gyeeta tree.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gysketch.h"
#include "../../include/gys_tdigest_tbl.h"

#define GYS_WAVE 64
// two constructs a host compiler cannot take as they stand; tests/cpp/kemu (the CPU stand-in of the device model that runs kernel
// LOGIC under g++) defines them its own way before this header is read
#ifndef GYS_OPAQUE_VGPR
#define GYS_OPAQUE_VGPR(x) asm volatile("" : "+v"(x)) // the value passes through an opaque move: nothing derived from it is loop-invariant
#endif
#ifndef GYS_DYN_LDS
#define GYS_DYN_LDS(type, name) extern __shared__ type name[] // the launch's dynamic LDS
#endif
#define GYS_SEED 0xceedfeadu   // common/gy_common_inc.h:1112 (every reference key hash uses this initval)
#define GYS_GOLDEN 0x9e3779b9u // common/jhash.h:37
#define GYS_NOSLOT 0xFFFFFFFFu
#define GYS_EMPTY_KEY 0xFFFFFFFFFFFFFFFFull

// A kernel's parameter block read from the kernel-argument segment where it is used (round 6, r6aa): a struct passed by value is loaded
// into SGPRs once and held -- and spilled into VGPR lanes -- across the whole kernel.  GYS_KERNARG_REF(T, p, p_arg) makes `p` a reference to
// the block in the kernel-argument segment (the kernel's ONLY parameter: offset 0); GYS_KERNARG_RELOAD(p) -- in uniform control flow only --
// forgets what was loaded, so that the uses behind it are scalar loads of their own.
#ifndef GYS_KERNARG
#define GYS_KERNARG 1
#endif
#if GYS_KERNARG && defined(__HIP_DEVICE_COMPILE__)
#define GYS_KERNARG_REF(T, name, arg)                                                             \
	typedef const T __attribute__((address_space(4))) *name##_kernarg_t;                      \
	name##_kernarg_t name##_k = (name##_kernarg_t)__builtin_amdgcn_kernarg_segment_ptr();     \
	const T &name = *(const T *)name##_k
#define GYS_KERNARG_RELOAD(name) asm volatile("" : "+s"(name##_k))
#else
#define GYS_KERNARG_REF(T, name, arg) const T &name = arg
#define GYS_KERNARG_RELOAD(name) (void)0
#endif

namespace gys {

// ------------------------------------------------------------------------------------------------ jhash (common/jhash.h)
__host__ __device__ __forceinline__ void jmix(uint32_t &a, uint32_t &b, uint32_t &c)
{
	// __jhash_mix common/jhash.h:23-34
	a -= b; a -= c; a ^= (c >> 13);
	b -= c; b -= a; b ^= (a << 8);
	c -= a; c -= b; c ^= (b >> 13);
	a -= b; a -= c; a ^= (c >> 12);
	b -= c; b -= a; b ^= (a << 16);
	c -= a; c -= b; c ^= (b >> 5);
	a -= b; a -= c; a ^= (c >> 3);
	b -= c; b -= a; b ^= (a << 10);
	c -= a; c -= b; c ^= (b >> 15);
}

// jhash_3words / jhash_2words common/jhash.h:122-135
__host__ __device__ __forceinline__ uint32_t jhash_3words(uint32_t a, uint32_t b, uint32_t c, uint32_t initval)
{
	a += GYS_GOLDEN;
	b += GYS_GOLDEN;
	c += initval;
	jmix(a, b, c);
	return c;
}

// get_uint64_hash common/gy_common_inc.h:1120-1123
__host__ __device__ __forceinline__ uint32_t get_uint64_hash(uint64_t k)
{
	return jhash_3words((uint32_t)(k & 0xFFFFFFFFu), (uint32_t)(k >> 32), 0, GYS_SEED);
}

// jhash2 common/jhash.h:88-113 over a small register array; N is the compile-time capacity, n the live word count
template <int N>
__host__ __device__ __forceinline__ uint32_t jhash2(const uint32_t (&k)[N], uint32_t n, uint32_t initval)
{
	uint32_t a = GYS_GOLDEN, b = GYS_GOLDEN, c = initval;
	uint32_t i = 0, len = n;
#pragma unroll
	for (int it = 0; it < N / 3; ++it) {
		if (len >= 3) {
			a += k[i];
			b += k[i + 1];
			c += k[i + 2];
			jmix(a, b, c);
			i += 3;
			len -= 3;
		}
	}
	c += n * 4;
	if (len == 2) {
		b += k[i + 1];
		a += k[i];
	} else if (len == 1) {
		a += k[i];
	}
	jmix(a, b, c);
	return c;
}

// jhash2 over exactly four words (the PAIR_IP_PORT key of an IPv4 flow: cli ip, cli port, ser ip, ser port): one full round + tail
__host__ __device__ __forceinline__ uint32_t jhash2_4w(uint32_t k0, uint32_t k1, uint32_t k2, uint32_t k3, uint32_t initval)
{
	uint32_t a = GYS_GOLDEN + k0, b = GYS_GOLDEN + k1, c = initval + k2;
	jmix(a, b, c);
	c += 16u; // length in bytes (common/jhash.h:103)
	a += k3;
	jmix(a, b, c);
	return c;
}

// two-word key (a 64-bit glob_id as lo,hi) through jhash2 with an arbitrary seed
__host__ __device__ __forceinline__ uint32_t jhash2_u64(uint64_t key, uint32_t initval)
{
	uint32_t a = GYS_GOLDEN, b = GYS_GOLDEN, c = initval;
	c += 8;
	b += (uint32_t)(key >> 32);
	a += (uint32_t)(key & 0xFFFFFFFFu);
	jmix(a, b, c);
	return c;
}

// slot hash of the per-host listener sub-tables (an engine-internal structure: k_resp_host probes it once per event).  Round 3: ONE 32-bit
// multiply per event -- the port enters through a 24-bit multiply (full rate on CDNA; a 32 x 32 multiply is a quarter-rate instruction and
// the earlier 64-bit multiplicative hash took three of them); the high half is folded down before the multiply (name spaces that differ
// in their high bits only) and the product's high bits onto its low bits after it.  Checked on arithmetic progressions of ports / name
// spaces, grids and random keys: probe lengths of a random function at both load factors.  The part of a many-listener host comes from
// bits 21.. of the product (the fold keeps the slots of one part spread over its whole table).
__host__ __device__ __forceinline__ uint32_t host_tbl_hash(uint32_t netns, uint32_t port)
{
	uint32_t x = netns ^ ((port & 0xFFFFu) * 0x9E3779u);
	x ^= x >> 16;
	return x * 0x85EBCA6Bu;
}
__host__ __device__ __forceinline__ uint32_t host_tbl_hash(uint64_t key48) { return host_tbl_hash((uint32_t)(key48 >> 16), (uint32_t)(key48 & 0xFFFFu)); }
__host__ __device__ __forceinline__ uint32_t host_tbl_slot(uint32_t hk, uint32_t mask) { return (hk ^ (hk >> 13)) & mask; }
__host__ __device__ __forceinline__ uint32_t host_tbl_part(uint32_t hk, uint32_t pmask) { return (hk >> 21) & pmask; }

// ------------------------------------------------------------------------------------------------ bucket hashes
struct HashDef {
	int32_t nthr;
	int32_t is_fixed_diff;
	int32_t arg_bits, t_bits;
	int64_t thr[14];
	int64_t fd_min, fd_maxp1, fd_diff;
};

// common/gy_statistics.h:1674-2063 + FIXED_DIFF_HASH :1584 (PERCENT_HASH): one table, a __constant__ copy for kernels and a
// host copy for the query path
#define GYS_HASH_DEFS_INIT                                                                                                  \
	{                                                                                                                   \
		{13, 0, 64, 64, {1, 10, 30, 60, 100, 150, 200, 300, 450, 700, 1000, 3000, 15000}, 0, 0, 0},                      \
		{12, 0, 32, 32, {1, 10, 100, 500, 1000, 5000, 25000, 50000, 100000, 300000, 1000000, 5000000}, 0, 0, 0},      \
		{13, 0, 32, 32, {1, 10, 50, 200, 500, 1000, 3000, 6000, 10000, 15000, 25000, 60000, 150000}, 0, 0, 0},        \
		{13, 0, 32, 32, {1, 10, 25, 50, 125, 400, 1000, 3000, 6000, 10000, 25000, 40000, 65000}, 0, 0, 0},            \
		{12, 0, 32, 32, {10, 25, 50, 75, 100, 150, 300, 500, 800, 1000, 2000, 5000}, 0, 0, 0},                          \
		{10, 0, 32, 32, {5, 10, 20, 40, 60, 80, 100, 140, 200, 250}, 0, 0, 0},                                          \
		{12, 0, 32, 32, {1, 5, 10, 25, 50, 75, 100, 150, 300, 500, 1000, 3000}, 0, 0, 0},                               \
		{11, 1, 64, 32, {9, 19, 29, 39, 49, 59, 69, 79, 89, 99, 100}, 0, 101, 10},                                      \
	}

__constant__ HashDef d_hash_defs[GYS_NUM_HASH_KINDS] = GYS_HASH_DEFS_INIT;
static const HashDef h_hash_defs[GYS_NUM_HASH_KINDS] = GYS_HASH_DEFS_INIT;

__host__ __device__ inline const HashDef &hash_def(int kind)
{
#ifdef __HIP_DEVICE_COMPILE__
	return d_hash_defs[kind];
#else
	return h_hash_defs[kind];
#endif
}

// RESP_TIME_HASH::get_bucket_from_data common/gy_statistics.h:1698-1725, specialised: the mid-slot shortcut + linear walk
// returns 1 + #{thresholds < data}; written as a branch-free compare sum (13 compares, no divergence inside a wave).
__device__ __forceinline__ uint32_t resp_bucket(int64_t data)
{
	if (data < 0) return 0;
	if (data >= 15001) return 14;
	const int32_t d = (int32_t)data;
	return 1u + (d > 1) + (d > 10) + (d > 30) + (d > 60) + (d > 100) + (d > 150) + (d > 200) + (d > 300) + (d > 450) + (d > 700) +
	       (d > 1000) + (d > 3000) + (d > 15000);
}

// generic table walker for every other hash class (same algorithm, different table)
__host__ __device__ inline uint32_t bucket_of(const HashDef &d, int64_t data)
{
	// GY_HISTOGRAM<T,..>::add_data(T data) narrows to T, then the hash's parameter type narrows again (int for most classes)
	if (d.t_bits == 32) data = (int64_t)(int32_t)data;
	if (d.arg_bits == 32) data = (int64_t)(int32_t)data;
	const int nb = d.nthr + 2;
	if (d.is_fixed_diff) {
		if (data < d.fd_min) return 0;
		if (data >= d.fd_maxp1) return (uint32_t)(nb - 1);
		return (uint32_t)(1 + (data - d.fd_min) / d.fd_diff);
	}
	if (data < 0) return 0;
	if (data >= d.thr[d.nthr - 1] + 1) return (uint32_t)(nb - 1);
	uint32_t b = 1;
	for (int i = 0; i < d.nthr; ++i) b += (data > d.thr[i]);
	return b;
}

// get_bucket_max_threshold<HashClass,T> common/gy_statistics.h:500-515
__host__ __device__ inline int64_t bucket_max_threshold(const HashDef &d, uint32_t id)
{
	const uint32_t nb = (uint32_t)d.nthr + 2;
	const int64_t min_value = d.is_fixed_diff ? d.fd_min : 0;
	const int64_t max_value = d.is_fixed_diff ? d.fd_maxp1 : d.thr[d.nthr - 1] + 1;
	if (id == 0) return d.t_bits == 32 ? (int64_t)(int32_t)(min_value - 1) : min_value - 1;
	if (id >= nb - 1) {
		const int64_t maxt = d.t_bits == 64 ? INT64_MAX : INT32_MAX;
		const int64_t lesst = max_value >= INT32_MAX ? INT64_MAX : (max_value > (INT16_MAX >> 1) ? INT32_MAX : INT16_MAX);
		return lesst < maxt ? lesst : maxt;
	}
	return d.thr[id - 1];
}

// GY_HISTOGRAM::get_percentiles common/gy_statistics.h:753-790 for ONE percentile on a 256-byte record
__host__ __device__ inline void hist_percentile(const HashDef &d, const gys_hist_rec &h, float pct, int64_t *data_value, int64_t *psum,
						uint64_t *pcount)
{
	const int nb = d.nthr + 2;
	const float multiplier = (float)((double)pct / 100.0); // float/double -> double, stored to a float (:757)
	const float prod = (float)h.total_count * multiplier;  // size_t * float (:758); plain mul, nothing to contract
	const uint64_t ncutoff = (uint64_t)prod;
	uint64_t total = 0;
	int64_t sum = 0;
	int i;
	for (i = 0; i < nb; ++i) {
		total += h.stats[i].count;
		sum += h.stats[i].sum;
		if (total >= ncutoff) break;
	}
	*pcount = total;
	*psum = sum;
	if (i < nb)
		*data_value = bucket_max_threshold(d, (uint32_t)i);
	else
		*data_value = bucket_max_threshold(d, h.total_count > 0 ? (uint32_t)nb : 0u); // :779-789
}

// SlabHistogramBuckets::getPercentileBucketIdx (thirdparty/SlabHistogramBucket.h:165-240), the rule TIME_HISTOGRAM::get_stats uses
// on a time level (common/gy_statistics.h:1352): first non-empty bucket whose cumulative fraction reaches pct01; empty -> bucket 1.
__host__ __device__ inline uint32_t slab_percentile_idx(const gys_hist_rec &h, int nb, double pct01)
{
	uint64_t total = 0, cur = 0;
	for (int i = 0; i < nb; ++i) total += h.stats[i].count;
	if (total == 0) return 1u;
	int idx;
	for (idx = 0; idx < nb; ++idx) {
		if (h.stats[idx].count == 0) continue;
		cur += h.stats[idx].count;
		if (pct01 <= (double)cur / (double)total) break;
	}
	return (uint32_t)idx;
}

// one TIME_HIST_VAL of TIME_HISTOGRAM::get_stats: ceiling of that bucket, negatives clamped to 0 (:1352-1354)
__host__ __device__ inline int64_t level_percentile(const HashDef &d, const gys_hist_rec &h, float pct)
{
	const int64_t v = bucket_max_threshold(d, slab_percentile_idx(h, d.nthr + 2, (double)pct / 100.0));
	return v < 0 ? 0 : v;
}

// ------------------------------------------------------------------------------------------------ open-addressing key table
struct TblEnt {
	uint64_t key; // GYS_EMPTY_KEY = free
	uint32_t val;
	uint32_t pad;
};

struct DevTable {
	TblEnt *ent; // 16-byte entries: one 16-B load per probe
	uint32_t mask;
};

// ---- wave-wide (64 lanes) scans and reductions on the DPP path.  __shfl_up / __shfl_xor compile to ds_bpermute_b32 -- an LDS instruction
// with ~100 cycles of latency, so the usual six-step loop is a 600-cycle dependent chain -- the DPP forms are VALU moves between lanes
// (row_shr 1 / 2 / 4 / 8 inside the rows of 16, then row_bcast:15 and row_bcast:31 across the rows: the sequence LLVM's own atomic
// optimizer emits for gfx9).  Only where all 64 lanes are active (every call site sits in wave-uniform code).  A host build (the g++ shim,
// tests/cpp/kemu) takes the shuffle loop.
#define GYS_DPP_STEP(x, op, ident)                                                                             \
	do {                                                                                                   \
		x = op(x, __builtin_amdgcn_update_dpp((int)(ident), x, 0x111, 0xf, 0xf, false)); /* row_shr:1 */   \
		x = op(x, __builtin_amdgcn_update_dpp((int)(ident), x, 0x112, 0xf, 0xf, false)); /* row_shr:2 */   \
		x = op(x, __builtin_amdgcn_update_dpp((int)(ident), x, 0x114, 0xf, 0xf, false)); /* row_shr:4 */   \
		x = op(x, __builtin_amdgcn_update_dpp((int)(ident), x, 0x118, 0xf, 0xf, false)); /* row_shr:8 */   \
		x = op(x, __builtin_amdgcn_update_dpp((int)(ident), x, 0x142, 0xa, 0xf, false)); /* row_bcast:15 -> rows 1, 3 */ \
		x = op(x, __builtin_amdgcn_update_dpp((int)(ident), x, 0x143, 0xc, 0xf, false)); /* row_bcast:31 -> rows 2, 3 */ \
	} while (0)
#define GYS_DPP_ADD(a, b) ((int)((uint32_t)(a) + (uint32_t)(b)))
#define GYS_DPP_UMIN(a, b) ((int)min((uint32_t)(a), (uint32_t)(b)))
#define GYS_DPP_UMAX(a, b) ((int)max((uint32_t)(a), (uint32_t)(b)))
#define GYS_DPP_SMIN(a, b) (min((int)(a), (int)(b)))
#define GYS_DPP_SMAX(a, b) (max((int)(a), (int)(b)))

// inclusive prefix sum over the lanes of the wave
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v)
{
#ifdef __HIP_DEVICE_COMPILE__
	int x = (int)v;
	GYS_DPP_STEP(x, GYS_DPP_ADD, 0);
	return (uint32_t)x;
#else
	const uint32_t lane = threadIdx.x & 63u;
	for (int d = 1; d < 64; d <<= 1) {
		const uint32_t t = __shfl_up(v, d, 64);
		if ((int)lane >= d) v += t;
	}
	return v;
#endif
}
// ... of 64-bit values: both halves travel by DPP, the add is a 64-bit one
__device__ __forceinline__ uint64_t wave_incl_scan_u64(uint64_t v)
{
#ifdef __HIP_DEVICE_COMPILE__
#define GYS_DPP64(ctrl, rmask)                                                                                     \
	do {                                                                                                       \
		const uint32_t lo_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, ctrl, rmask, 0xf, false);         \
		const uint32_t hi_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), ctrl, rmask, 0xf, false); \
		v += (uint64_t)lo_ | ((uint64_t)hi_ << 32);                                                        \
	} while (0)
	GYS_DPP64(0x111, 0xf);
	GYS_DPP64(0x112, 0xf);
	GYS_DPP64(0x114, 0xf);
	GYS_DPP64(0x118, 0xf);
	GYS_DPP64(0x142, 0xa);
	GYS_DPP64(0x143, 0xc);
#undef GYS_DPP64
	return v;
#else
	const uint32_t lane = threadIdx.x & 63u;
	for (int d = 1; d < 64; d <<= 1) {
		const uint64_t t = __shfl_up(v, d, 64);
		if ((int)lane >= d) v += t;
	}
	return v;
#endif
}
// the value of lane 63 (after wave_incl_scan_u32: the wave's total), the same in every lane
__device__ __forceinline__ uint32_t wave_last_u32(uint32_t v)
{
#ifdef __HIP_DEVICE_COMPILE__
	return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
#else
	return (uint32_t)__shfl((int)v, 63, 64);
#endif
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) { return wave_last_u32(wave_incl_scan_u32(v)); }
#ifdef __HIP_DEVICE_COMPILE__
#define GYS_WAVE_REDUCE(name, type, op, ident)                          \
	__device__ __forceinline__ type name(type v)                    \
	{                                                               \
		int x = (int)v;                                         \
		GYS_DPP_STEP(x, op, ident);                             \
		return (type)__builtin_amdgcn_readlane(x, 63);          \
	}
#else
#define GYS_WAVE_REDUCE(name, type, op, ident)                                              \
	__device__ __forceinline__ type name(type v)                                        \
	{                                                                                   \
		for (int d = 32; d >= 1; d >>= 1) v = (type)op(v, __shfl_xor(v, d, 64));    \
		return v;                                                                   \
	}
#endif
GYS_WAVE_REDUCE(wave_min_u32, uint32_t, GYS_DPP_UMIN, 0xFFFFFFFFu)
GYS_WAVE_REDUCE(wave_max_u32, uint32_t, GYS_DPP_UMAX, 0u)
GYS_WAVE_REDUCE(wave_min_i32, int32_t, GYS_DPP_SMIN, 0x7FFFFFFF)
GYS_WAVE_REDUCE(wave_max_i32, int32_t, GYS_DPP_SMAX, (int)0x80000000)

// probe hash = the reference's own 64-bit-id hash (get_uint64_hash) so bucket choice mirrors listen_tbl_ lookups
// (server/gy_mconnhdlr.cc:11180-11183)
__device__ __forceinline__ uint32_t tbl_lookup(const DevTable &t, uint64_t key)
{
	if (key == GYS_EMPTY_KEY) return GYS_NOSLOT; // (never registered: gys_register_listeners refuses it -- and it would match a free entry below)
	uint32_t h = get_uint64_hash(key) & t.mask;
	for (uint32_t probes = 0; probes <= t.mask; ++probes) {
		const uint4 e = *(const uint4 *)&t.ent[h];
		const uint64_t k = (uint64_t)e.x | ((uint64_t)e.y << 32);
		if (k == key) return e.z;
		if (k == GYS_EMPTY_KEY) return GYS_NOSLOT;
		h = (h + 1) & t.mask;
	}
	return GYS_NOSLOT;
}

// listener tuple key: (host_slot:16 | netns:32 | port:16).  The reference looks a response event's listener up per host by
// NS_IP_PORT ignoring the IP (ANY_IP, common/gy_inet_inc.h:160-172 used at common/gy_socket_stat.cc:1671).
__host__ __device__ __forceinline__ uint64_t listener_key(uint32_t host_slot, uint32_t netns, uint16_t port)
{
	return ((uint64_t)host_slot << 48) | ((uint64_t)netns << 16) | (uint64_t)port;
}

// ------------------------------------------------------------------------------------------------ sketches
// 64-bit sketch hash = (jhash2(seed 0xceedfead) << 32) | jhash2(seed 0x9e3779b9)   
template <int N>
__host__ __device__ __forceinline__ uint64_t hash64(const uint32_t (&w)[N], uint32_t n)
{
	return ((uint64_t)jhash2<N>(w, n, GYS_SEED) << 32) | (uint64_t)jhash2<N>(w, n, GYS_GOLDEN);
}

__host__ __device__ __forceinline__ void hll_idx_rank(uint64_t h, int p, uint32_t *idx, uint32_t *rank)
{
	const uint64_t w = h << p;
	*idx = (uint32_t)(h >> (64 - p));
#ifdef __HIP_DEVICE_COMPILE__
	*rank = w ? (uint32_t)__clzll((long long)w) + 1u : (uint32_t)(64 - p + 1);
#else
	*rank = w ? (uint32_t)__builtin_clzll(w) + 1u : (uint32_t)(64 - p + 1);
#endif
}

// PAIR_IP_PORT::get_hash key bytes (common/gy_inet_inc.h:225-247) from two IPv4-or-IPv6 endpoints:
// [cli inaddr][cli port LE, 00 00][ser inaddr][ser port LE, 00 00]; inaddr = 4 bytes if ip32_be != 0 else the 16 ip128 bytes
// (GY_IP_ADDR::get_as_inaddr common/gy_common_inc.h:10950-10959).  Returns the word count (4, 7 or 10).
__host__ __device__ __forceinline__ uint32_t pair_words(uint32_t cip32, const uint32_t cip128[4], uint16_t cport, uint32_t sip32,
							const uint32_t sip128[4], uint16_t sport, uint32_t (&w)[10])
{
	uint32_t n = 0;
	if (cip32) {
		w[n++] = cip32;
	} else {
		w[n++] = cip128[0]; w[n++] = cip128[1]; w[n++] = cip128[2]; w[n++] = cip128[3];
	}
	w[n++] = (uint32_t)cport;
	if (sip32) {
		w[n++] = sip32;
	} else {
		w[n++] = sip128[0]; w[n++] = sip128[1]; w[n++] = sip128[2]; w[n++] = sip128[3];
	}
	w[n++] = (uint32_t)sport;
	for (uint32_t i = n; i < 10; ++i) w[i] = 0;
	return n;
}

// ------------------------------------------------------------------------------------------------ listener addresses
// GY_IP_ADDR::set_ip(unsigned __int128) (common/gy_common_inc.h:10686-10692) zeroes ip32_be_ and then calls get_ipv6_type_flags
// (:11040-11129), which stores the IPv4 address an IPv6 address EMBEDS into embedded_ipv4_ -- the same storage as ip32_be_ (the union at
// :10497-10500).  An address of 2002::/16 (6to4: bytes 2..5), ::ffff:a.b.c.d (mapped: bytes 12..15) or 64:ff9b::/32 (NAT64: bytes 12..15)
// therefore ends with ip32_be_ = the embedded address, hashes as those 4 bytes (get_as_inaddr :10950-10959) and compares equal to that
// IPv4 address (operator== :10629-10636).  a = the 16 address bytes as four words loaded in memory order.  Returns that ip32_be_ (0: none).
__host__ __device__ __forceinline__ uint32_t ip6_embedded_v4(const uint32_t (&a)[4])
{
	if ((a[0] | a[1] | a[2] | a[3]) == 0u) return 0u;                    // :: (IPv6_ANY)
	if ((a[0] | a[1] | a[2]) == 0u && a[3] == 0x01000000u) return 0u;    // ::1
	if ((a[0] & 0xF0u) == 0x20u)                                         // 2000::/4: only 2002::/16 carries an address
		return (a[0] & 0xFFFFu) == 0x0220u ? ((a[0] >> 16) | (a[1] << 16)) : 0u;
	if ((a[0] | a[1]) == 0u && a[2] == 0xFFFF0000u) return a[3];         // ::ffff:a.b.c.d
	if (a[0] == 0x9BFF6400u) return a[3];                                // 64:ff9b::/32 (the check at :11098 reads the first four bytes only)
	return 0u;
}

// one listener of a (netns, port) key that needs more than the key to be told apart: bound to an address, or not alone on its key.
// operator==(shared_ptr<TCP_LISTENER>, NS_IP_PORT) (common/gy_socket_stat.h:708-714): inode and port equal (the table key here) and
// is_any_ip_ || listener address == event address, the addresses compared as GY_IP_ADDR::operator== does (:10629-10636: by ip32_be_ when
// either side has one, else by the 16 bytes).  The candidates of a key stand in registration order and the first match wins (the
// reference's lookup_single_elem walks the hash chain from its oldest entry: liburcu's _cds_lfht_add puts a node behind the nodes of
// equal hash that are already there -- that library is not part of the reference tree, its published behaviour is restated).
struct ListenerCand {
	uint32_t ip32;     // ip32_be_ of the listener's address (an IPv4 address, or the one its IPv6 address embeds)
	uint32_t flags;    // bit 0: is_any_ip_; bit 1: last candidate of the key
	uint32_t local;    // local index inside the host's (part's) sub-table
	uint32_t slot;     // service slot
	uint32_t ip128[4]; // ip128_be_ (zero for an IPv4 address)
};
static_assert(sizeof(ListenerCand) == 32, "two 16-byte loads per candidate");
#define GYS_LOCAL_GROUP 0x8000u      // sub-table entry: the low 15 bits index the host's candidate region instead of naming a local index
#define GYS_SLOT_GROUP 0x80000000u   // global listener table value: the low 31 bits index the candidate pool

__device__ __forceinline__ bool cand_resolve(const ListenerCand *cand, uint32_t idx, uint32_t e32, const uint32_t (&e128)[4], uint32_t *local, uint32_t *slot)
{
	for (;; ++idx) {
		const uint4 a = ((const uint4 *)cand)[2u * idx], b = ((const uint4 *)cand)[2u * idx + 1u];
		const bool same = (a.x | e32) ? a.x == e32 : (b.x == e128[0] && b.y == e128[1] && b.z == e128[2] && b.w == e128[3]);
		if ((a.y & 1u) || same) {
			*local = a.z;
			*slot = a.w;
			return true;
		}
		if (a.y & 2u) return false;
	}
}

// ------------------------------------------------------------------------------------------------ t-digest cluster thresholds
__constant__ uint64_t c_td_bnd[GYS_TDIGEST_NB + 1] = GYS_TDIGEST_BND_INIT;
static const uint64_t h_td_bnd[GYS_TDIGEST_NB + 1] = GYS_TDIGEST_BND_INIT;

// cluster(mid2) = #{ j in 1..NB-1 : BND[j] * twoN <= mid2 * 2^32 } = #{ j : mid2 >= T_j },  T_j = ceil(BND[j] * twoN / 2^32)
__device__ __forceinline__ uint64_t td_threshold(uint64_t bnd, uint64_t twoN)
{
	const uint64_t lo = bnd * twoN;
	const uint64_t hi = __umul64hi(bnd, twoN);
	uint64_t t = (hi << 32) | (lo >> 32);
	if (lo & 0xFFFFFFFFull) t += 1;
	return t;
}

} // namespace gys
