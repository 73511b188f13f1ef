// Roll-up digests: the response-time digest of a GROUP of services (a host, a cluster, all hosts of this rank, all ranks).
// Reference analogue: the aggregated percentile of a set of listeners is computed by Postgres from its members' rows,
// public.tdigest_percentile(col, 100, p) (common/gy_query_common.cc:1818-1855); the cluster-level fan-in is
// SHCONN_HANDLER::aggregate_cluster_state (server/gy_shconnhdlr.cc:4583-4720).  Definition (frozen in oracle/gy_oracle_rollup.c,
// gyo_tdbins_*; round 6 -- until then the roll-up was a left fold over the members, 10^7 sequential merge steps = 0.3 s per query):
//   THE UNION BY VALUE BIN.  The 2048 value bins of k_digest_bins (mb_bin: one bin per millisecond below 1024, 64 cells per octave
//   above) each hold the exact 64-bit {sum, count} of what the group's members bring: a member's non-empty clusters, whole, into the
//   bin of ceil(sum / count); a service's buffered values into the bin of the value.  Additions commute: the bins of a group do not
//   depend on the order or the grouping in which the members are visited -- any number of workgroups add to them.  Then the bins, in
//   order, are laid on the rank axis: bin b (weight w, W = weight of the bins below) occupies the unit mid-points 2 (W + r) + 1,
//   r < w; point r belongs to the cluster every merge of the engine would give that mid-point; the points r0 <= r < r1 of a bin that
//   fall into one cluster bring it floor(sum r1 / w) - floor(sum r0 / w) of the bin's sum (128-bit product).
// Two kernels: k_rollup_accum (HBM-bound: 12 bytes per cluster, 4 bytes per buffered value, one LDS atomic or two per item; a
// workgroup's LDS bins go to the group's bins in HBM with one 64-bit atomic pair per non-empty bin) and k_rollup_cluster (one
// workgroup per group, 32 KB in, one slab out).  A roll-up of roll-ups (hosts -> cluster / global, ranks -> all) is the same pair
// over the members' clusters.
#pragma once

namespace gys {

#define GYS_RB_STRIDE (2u * GYS_MB_BINS + 2u) // 64-bit words of a group's bins in HBM: cnt[2048], sum[2048], vmin, vmax
#ifndef GYS_RB_NT
#define GYS_RB_NT 1024u                       // threads of an accumulating workgroup: 16 waves, each walks members of its own (two workgroups per CU share its LDS)
#endif
#ifndef GYS_RB_AHEAD
#define GYS_RB_AHEAD 2u                       // 16-byte pieces per lane (1 KB per wave each) of a member's buffered values that are requested before its clusters are looked at
#endif
#ifndef GYS_RB_WAVES
#define GYS_RB_WAVES 8                        // waves per SIMD the accumulate kernel is compiled for (0: the compiler's choice).  The kernel lives on waves in flight: 8 (64 VGPRs, four of them spilled) 13.9 ms, 6 (80 VGPRs) 17.0 ms, 3 - 4.5 28 - 30 ms (profiles/r6ag_*)
#endif
#ifndef GYS_RB_VC
#define GYS_RB_VC 4096u                       // buffered values below this are COUNTED per exact value in LDS (one 32-bit add each: their sums follow from the counts)
#endif

struct RollupChunk {
	uint32_t group, m0, m1, pad; // members[m0, m1) belong to `group`
};

struct RollupP {
	DigestP d;
	const RollupChunk *chunks;
	uint32_t nchunks;
	const uint32_t *members; // kind 0: service slots; kind 1: indices into `in`
	int kind;
	const gys_tdigest_slab *in;
	unsigned long long *bins; // [ngroups][GYS_RB_STRIDE]
	gys_tdigest_slab *out;    // [ngroups]
	uint32_t ngroups;
};

__global__ __launch_bounds__(256) void k_rollup_init(unsigned long long *bins, uint32_t ngroups)
{
	const size_t n = (size_t)ngroups * GYS_RB_STRIDE;
	for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (size_t)gridDim.x * 256u) {
		const uint32_t k = (uint32_t)(i % GYS_RB_STRIDE);
		bins[i] = k == 2u * GYS_MB_BINS ? (unsigned long long)(long long)INT32_MAX : k == 2u * GYS_MB_BINS + 1u ? (unsigned long long)(long long)INT32_MIN : 0ull;
	}
}

// ceil(sum / cnt) for 0 < sum < 2^63, cnt != 0, when the quotient is below 2^32 (else: ~0): the double quotient is off by at most one
__device__ __forceinline__ uint32_t ceil_div_wide(uint64_t sum, uint64_t cnt)
{
	if ((int64_t)sum <= 0) return 0u; // (not an engine digest: a caller's slab; the oracle's rule)
	const double dq = (double)sum / (double)cnt;
	if (dq >= 4294967040.0) return 0xFFFFFFFFu;
	uint64_t f = (uint64_t)dq;
	int64_t r = (int64_t)(sum - f * cnt);
	if (r < 0) {
		f--;
		r += (int64_t)cnt;
	} else if ((uint64_t)r >= cnt) {
		f++;
		r -= (int64_t)cnt;
	}
	f += r != 0;
	return f > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)f;
}

__device__ __forceinline__ uint32_t rb_bin(uint32_t v) { return mb_bin(v < (1u << 26) ? v : (1u << 26) - 1u); } // (a staged word carries 26 value bits)

#if GYS_RB_WAVES
__global__ __launch_bounds__(GYS_RB_NT, GYS_RB_WAVES) void k_rollup_accum(RollupP q)
#else
__global__ __launch_bounds__(GYS_RB_NT) void k_rollup_accum(RollupP q)
#endif
{
	const DigestP &p = q.d;
	__shared__ unsigned long long s_cnt[GYS_MB_BINS], s_sum[GYS_MB_BINS]; // clusters (any bin) and buffered values >= 1024
	// buffered values below GYS_RB_VC: one counter per exact value -- the count says it all, and minimum / maximum of the values come from the
	// counters as well: a value is a compare, two address instructions and one LDS add.  Larger values (0.2 % of the bench's stream) take the
	// general path: cell number, two 64-bit adds, own minimum / maximum.  (Measured on the way, profiles/r6z_*, r6ag_*: reading the member's
	// identity one member ahead and asking for its first 2 KB of values before its clusters 17.9 -> 17.5 ms; this value path instead of bin + two
	// adds + min / max per value 17.5 -> 16.8 ms; eight counters per value below 256, by lane -- in case lanes that hold the same value
	// serialize in the LDS -- 17.6 ms; 4 or 8 KB requested ahead 20 - 21 ms (registers: fewer waves); 12 - 20 waves per CU instead of 24: 28 - 30 ms;
	// 32 waves per CU: 13.9 ms -- the kernel is bound by the latency of its loads, i.e. by the number of waves that wait at once.)
	__shared__ uint32_t s_vc[GYS_RB_VC];
	__shared__ long long s_mm[2];
	const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
	constexpr uint32_t NW = GYS_RB_NT / 64u;

	for (uint32_t c = blockIdx.x; c < q.nchunks; c += gridDim.x) {
		for (uint32_t k = tid; k < GYS_MB_BINS; k += GYS_RB_NT) {
			s_cnt[k] = 0;
			s_sum[k] = 0;
		}
		for (uint32_t k = tid; k < GYS_RB_VC; k += GYS_RB_NT) s_vc[k] = 0;
		if (tid == 0) {
			s_mm[0] = INT32_MAX;
			s_mm[1] = INT32_MIN;
		}
		__syncthreads();
		const RollupChunk ck = q.chunks[c];
		long long mmin = INT32_MAX, mmax = INT32_MIN; // lane 0: the extremes of the members that bring clusters
		int32_t lmin = INT32_MAX, lmax = INT32_MIN;   // the buffered values'
		// a wave per member, no barrier inside a chunk.  The member's identity and buffer fill are read one member ahead and its first 2 KB of
		// buffered values are requested before its clusters: one round trip to HBM per member instead of three dependent ones
		uint32_t mem_n = 0, npend_n = 0;
		if (ck.m0 + wave < ck.m1) {
			mem_n = q.members[ck.m0 + wave];
			if (q.kind == 0) npend_n = min(p.td_meta[mem_n].npend, p.pend_cap); // (between batches a buffer holds at most pend_cap values)
		}
		for (uint32_t mi = ck.m0 + wave; mi < ck.m1; mi += NW) {
			const uint32_t mem = mem_n, npend = npend_n;
			if (mi + NW < ck.m1) {
				mem_n = q.members[mi + NW];
				if (q.kind == 0) npend_n = min(p.td_meta[mem_n].npend, p.pend_cap);
			}
			const uint32_t *pend = p.td_pend + (size_t)mem * p.pcap;
			const bool quads = q.kind == 0 && (p.pcap & 3u) == 0u; // 16 bytes per lane and request
			uint4 va[GYS_RB_AHEAD];
#pragma unroll
			for (uint32_t k = 0; k < GYS_RB_AHEAD; ++k) {
				va[k] = make_uint4(0, 0, 0, 0);
				if (quads && 4u * lane + 256u * k < npend) va[k] = ((const uint4 *)pend)[lane + 64u * k];
			}
			// ---- the member's clusters, whole, into the bin of the integer threshold of their mean
			bool any = false;
#pragma unroll
			for (uint32_t k = 0; k < (GYS_TD_NB + 63u) / 64u; ++k) {
				const uint32_t j = lane + 64u * k;
				uint64_t cn = 0, sm = 0;
				if (j < GYS_TD_NB) {
					if (q.kind == 0) {
						cn = p.td_cnt[(size_t)mem * GYS_TD_NB + j];
						sm = (uint64_t)p.td_sum[(size_t)mem * GYS_TD_NB + j];
					} else {
						cn = q.in[mem].cnt[j];
						sm = (uint64_t)q.in[mem].sum[j];
					}
				}
				if (cn) {
					const uint32_t b = rb_bin(ceil_div_wide(sm, cn));
					atomicAdd(&s_cnt[b], (unsigned long long)cn);
					atomicAdd(&s_sum[b], (unsigned long long)sm);
				}
				any |= __ballot(cn != 0) != 0ull;
			}
			if (any && lane == 0) { // the extremes of a member that brings clusters
				long long vmn, vmx;
				if (q.kind == 0) {
					const int2 mm = p.td_minmax[mem];
					vmn = mm.x;
					vmx = mm.y;
				} else {
					vmn = q.in[mem].vmin;
					vmx = q.in[mem].vmax;
				}
				mmin = vmn < mmin ? vmn : mmin;
				mmax = vmx > mmax ? vmx : mmax;
			}
			if (q.kind != 0) continue;
			// ---- a service's buffered values: unit points
			auto one = [&](uint32_t word) {
				if (word < (GYS_RB_VC << GYS_ROW_BITS)) {
					atomicAdd((uint32_t *)((char *)s_vc + ((word >> (GYS_ROW_BITS - 2u)) & ~3u)), 1u);
				} else {
					const uint32_t uv = word >> GYS_ROW_BITS, b = rb_bin(uv);
					atomicAdd(&s_cnt[b], 1ull);
					atomicAdd(&s_sum[b], (unsigned long long)uv);
					lmin = min(lmin, (int32_t)uv);
					lmax = max(lmax, (int32_t)uv);
				}
			};
			auto quad = [&](const uint4 w4, uint32_t i) {
				if (i + 3u < npend) { // (all but the buffer's last quad)
					one(w4.x);
					one(w4.y);
					one(w4.z);
					one(w4.w);
				} else {
					if (i < npend) one(w4.x);
					if (i + 1u < npend) one(w4.y);
					if (i + 2u < npend) one(w4.z);
				}
			};
			if (quads) {
#pragma unroll
				for (uint32_t k = 0; k < GYS_RB_AHEAD; ++k) quad(va[k], 4u * lane + 256u * k);
#pragma unroll 2
				for (uint32_t i = 4u * lane + 256u * GYS_RB_AHEAD; i < npend; i += 256u) quad(((const uint4 *)pend)[i >> 2], i);
			} else {
#pragma unroll 4
				for (uint32_t i = lane; i < npend; i += 64u) one(pend[i]);
			}
		}
		__syncthreads();
		// the counted values: those of 1024 and more join their cells; the smallest / largest counted value
		for (uint32_t k = tid; k < GYS_RB_VC; k += GYS_RB_NT) {
			const uint32_t vc = s_vc[k];
			if (!vc) continue;
			lmin = min(lmin, (int32_t)k);
			lmax = max(lmax, (int32_t)k);
			if (k >= GYS_MB_EXACT) {
				const uint32_t b = rb_bin(k);
				atomicAdd(&s_cnt[b], (unsigned long long)vc);
				atomicAdd(&s_sum[b], (unsigned long long)vc * k);
			}
		}
		lmin = wave_min_i32(lmin);
		lmax = wave_max_i32(lmax);
		if (lane == 0) {
			mmin = (long long)lmin < mmin ? (long long)lmin : mmin;
			mmax = (long long)lmax > mmax ? (long long)lmax : mmax;
			if (mmin != INT32_MAX) atomicMin(&s_mm[0], mmin);
			if (mmax != INT32_MIN) atomicMax(&s_mm[1], mmax);
		}
		__syncthreads();
		// ---- the workgroup's bins into the group's: one pair of 64-bit adds per non-empty bin
		unsigned long long *gb = q.bins + (size_t)ck.group * GYS_RB_STRIDE;
		for (uint32_t k = tid; k < GYS_MB_BINS; k += GYS_RB_NT) {
			unsigned long long cn = s_cnt[k], sm = s_sum[k];
			if (k < GYS_MB_EXACT) {
				const uint32_t vc = s_vc[k];
				cn += vc;
				sm += (unsigned long long)vc * k;
			}
			if (cn) {
				atomicAdd(&gb[k], cn);
				atomicAdd(&gb[GYS_MB_BINS + k], sm);
			}
		}
		if (tid == 0) {
			if (s_mm[0] != INT32_MAX) atomicMin((long long *)&gb[2u * GYS_MB_BINS], s_mm[0]);
			if (s_mm[1] != INT32_MIN) atomicMax((long long *)&gb[2u * GYS_MB_BINS + 1u], s_mm[1]);
		}
		__syncthreads();
	}
}

// floor(s k / w) for k < w (the quotient is below s: it fits)
__device__ __forceinline__ uint64_t mul_div_floor(uint64_t s, uint64_t k, uint64_t w)
{
	const uint64_t hi = __umul64hi(s, k), lo = s * k;
	if (!hi) return lo / w;
	uint64_t qt = 0, rem = hi; // hi < w
	for (int i = 63; i >= 0; --i) {
		const bool carry = (rem >> 63) != 0;
		rem = (rem << 1) | ((lo >> i) & 1ull);
		qt <<= 1;
		if (carry || rem >= w) {
			rem -= w;
			qt |= 1ull;
		}
	}
	return qt;
}

__device__ __forceinline__ uint32_t cluster_of_u64(const uint64_t *T, uint64_t mid2)
{
	uint32_t a = 0;
#pragma unroll
	for (uint32_t step = GYS_NBP / 2; step >= 1u; step >>= 1)
		if (mid2 >= T[a + step]) a += step;
	return a;
}

__global__ __launch_bounds__(256) void k_rollup_cluster(RollupP q)
{
	__shared__ uint64_t s_T[GYS_NBP];
	__shared__ unsigned long long o_sum[GYS_NBP], o_cnt[GYS_NBP];
	__shared__ uint64_t s_ww[4];
	const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;

	for (uint32_t g = blockIdx.x; g < q.ngroups; g += gridDim.x) {
		const unsigned long long *gb = q.bins + (size_t)g * GYS_RB_STRIDE;
		uint64_t cn[GYS_MB_BPT], sm[GYS_MB_BPT], own = 0; // thread t: bins [8 t, 8 t + 8)
#pragma unroll
		for (uint32_t k = 0; k < GYS_MB_BPT; ++k) {
			cn[k] = gb[GYS_MB_BPT * tid + k];
			sm[k] = gb[GYS_MB_BINS + GYS_MB_BPT * tid + k];
			own += cn[k];
		}
		const uint64_t inc = wave_incl_scan_u64(own);
		if (lane == 63u) s_ww[wave] = inc;
		o_sum[tid] = 0;
		o_cnt[tid] = 0;
		__syncthreads();
		uint64_t W = inc - own, N = 0;
#pragma unroll
		for (uint32_t k = 0; k < 4u; ++k) {
			if (k < wave) W += s_ww[k];
			N += s_ww[k];
		}
		s_T[tid] = (tid >= 1u && tid < GYS_TD_NB) ? td_threshold(c_td_bnd[tid], 2ull * N) : (tid ? ~0ull : 0ull);
		__syncthreads();
		if (N) {
#pragma unroll 1
			for (uint32_t k = 0; k < GYS_MB_BPT; ++k) {
				const uint64_t w = cn[k], s = sm[k];
				if (!w) continue;
				uint32_t a = cluster_of_u64(s_T, 2ull * W + 1ull);
				uint64_t r = 0, given = 0;
				while (r < w) { // (nearly always one round: a cluster spans far more ranks than a bin holds)
					const uint64_t Tn = s_T[a + 1u]; // first mid-point of the next cluster (~0 after the last)
					uint64_t r1 = w;
					if (Tn != ~0ull) {
						const uint64_t x = Tn > 2ull * W ? (Tn - 2ull * W) >> 1 : 0ull; // points with 2 (W + r) + 1 < Tn
						r1 = x < w ? x : w;
					}
					if (r1 > r) {
						const uint64_t upto = r1 == w ? s : mul_div_floor(s, r1, w);
						atomicAdd(&o_sum[a], (unsigned long long)(upto - given));
						atomicAdd(&o_cnt[a], (unsigned long long)(r1 - r));
						given = upto;
						r = r1;
					}
					++a;
				}
				W += w;
			}
		}
		__syncthreads();
		if (tid < GYS_TD_NB) {
			q.out[g].sum[tid] = (int64_t)o_sum[tid];
			q.out[g].cnt[tid] = o_cnt[tid];
		}
		if (tid == 0) {
			q.out[g].vmin = (int64_t)gb[2u * GYS_MB_BINS];
			q.out[g].vmax = (int64_t)gb[2u * GYS_MB_BINS + 1u];
		}
		__syncthreads();
	}
}

} // namespace gys
