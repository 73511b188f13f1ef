// Roll-up digests: the response-time digest of a GROUP of services (a host, a cluster, all hosts of this rank, all ranks).
// Reference analogue: the aggregated percentile of a set of listeners is computed by Postgres from its members' rows,
// public.tdigest_percentile(col, 100, p) (common/gy_query_common.cc:1818-1855); the cluster-level fan-in is
// SHCONN_HANDLER::aggregate_cluster_state (server/gy_shconnhdlr.cc:4583-4720).  Definition (frozen in oracle/gy_oracle_rollup.c):
//   rollup(group) = left fold over the members in order of  d := merge(d, member);
//   a service contributes its clusters (weighted points at their means) and then its buffered values (unit points);
//   a roll-up digest contributes its clusters; merge = the exact-integer k-bucket merge of the per-service digests with 64-bit
//   counters (a group's weight passes 2^32 within a few windows).
// One 256-thread workgroup per group walks its members one after the other (the fold is sequential by definition); inside a
// member step the work is data-parallel: both cluster lists are sorted by mean, so the cross ranks come from binary searches with
// exact 128-bit rational compares, and the buffered values use the value-bin counting of k_digest_bins.  Query-time code: a host's
// 1 000 services take a few ms per workgroup, 10^4 hosts ~40 ms on the whole chip.
#pragma once

namespace gys {

struct RollupP {
	DigestP d;
	const uint32_t *off;      // [ngroups + 1]
	const uint32_t *members;  // kind 0: service slots; kind 1: indices into `in`
	int kind;
	const gys_tdigest_slab *in;
	gys_tdigest_slab *out;    // [ngroups]
	uint32_t ngroups;
};

// a * b < c * d  (all < 2^64, exact)
__device__ __forceinline__ bool mul_lt(uint64_t a, uint64_t b, uint64_t c, uint64_t d)
{
	const uint64_t h1 = __umul64hi(a, b), h2 = __umul64hi(c, d);
	return h1 != h2 ? h1 < h2 : a * b < c * d;
}
__device__ __forceinline__ bool mul_le(uint64_t a, uint64_t b, uint64_t c, uint64_t d) { return !mul_lt(c, d, a, b); }

// block-wide (256 threads) order-preserving compaction of the entries with cnt != 0 into (c_sum, c_cnt) and the exclusive prefix of
// their weights c_wpfx[0..n]; returns n, *total = weight of all.  Every thread calls it with ITS entry (index = thread).
__device__ __forceinline__ uint32_t compact_256(int64_t sum, uint64_t cnt, int64_t *c_sum, uint64_t *c_cnt, uint64_t *c_wpfx, uint64_t *total,
						uint32_t *s_wv, uint64_t *s_ww)
{
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	const unsigned long long b = __ballot(cnt != 0);
	uint64_t inc = cnt;
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		const uint64_t t = __shfl_up(inc, d, 64);
		if ((int)lane >= d) inc += t;
	}
	__syncthreads(); // the previous use of s_wv / s_ww is over
	if (lane == 63u) s_ww[wave] = inc;
	if (lane == 0u) s_wv[wave] = (uint32_t)__popcll(b);
	__syncthreads();
	uint32_t pb = 0, n = 0;
	uint64_t wb = 0, tot = 0;
#pragma unroll
	for (uint32_t k = 0; k < 4u; ++k) {
		if (k < wave) {
			pb += s_wv[k];
			wb += s_ww[k];
		}
		n += s_wv[k];
		tot += s_ww[k];
	}
	if (cnt) {
		const uint32_t pos = pb + (uint32_t)__popcll(b & (lane ? (~0ull >> (64 - lane)) : 0ull));
		c_sum[pos] = sum;
		c_cnt[pos] = cnt;
		c_wpfx[pos] = wb + inc - cnt;
	}
	if (threadIdx.x == 0) c_wpfx[n] = tot;
	*total = tot;
	__syncthreads();
	return n;
}

__device__ __forceinline__ uint32_t cluster_of_u64(const uint64_t *T, uint64_t mid2)
{
	uint32_t a = 0;
#pragma unroll
	for (uint32_t step = GYS_NBP / 2; step >= 1u; step >>= 1)
		if (mid2 >= T[a + step]) a += step;
	return a;
}

// VPT: buffered values of a member per thread (4: buffers of up to 1024 values -- the default td_pend_cap; 8 / 16 for larger ones)
template <uint32_t VPT = 4u>
__global__ __launch_bounds__(256) void k_digest_rollup(RollupP q)
{
	const DigestP &p = q.d;
	__shared__ int64_t d_sum[GYS_NBP], c_sum[GYS_NBP], n_sum[GYS_NBP];
	__shared__ uint64_t d_cnt[GYS_NBP], c_cnt[GYS_NBP], n_cnt[GYS_NBP], c_wpfx[GYS_NBP + 1], n_wpfx[GYS_NBP + 1];
	__shared__ uint64_t s_T[GYS_NBP];
	__shared__ unsigned long long o_sum[GYS_NBP], o_cnt[GYS_NBP];
	__shared__ __align__(16) uint32_t s_bin[GYS_MB_BINS];
	__shared__ uint32_t s_big[256u * VPT];
	__shared__ uint32_t s_thr[GYS_NBP];
	__shared__ uint32_t s_wv[4], s_ws[4], s_nbig;
	__shared__ uint64_t s_ww[4];
	__shared__ long long s_mm[2];
	const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;

	for (uint32_t g = blockIdx.x; g < q.ngroups; g += gridDim.x) {
		d_sum[tid] = 0;
		d_cnt[tid] = 0;
		if (tid == 0) {
			s_mm[0] = INT32_MAX;
			s_mm[1] = INT32_MIN;
		}
		__syncthreads();
		const uint32_t m0 = q.off[g], m1 = q.off[g + 1];
		for (uint32_t mi = m0; mi < m1; ++mi) {
			const uint32_t mem = q.members[mi];
			// ---------------- step A: the member's clusters as weighted points
			int64_t ns = 0;
			uint64_t ncn = 0;
			uint32_t npend = 0;
			if (q.kind == 0) {
				if (tid < GYS_TD_NB) {
					ncn = p.td_cnt[(size_t)mem * GYS_TD_NB + tid];
					ns = p.td_sum[(size_t)mem * GYS_TD_NB + tid];
				}
				npend = min(p.td_meta[mem].npend, 256u * VPT);
			} else if (tid < GYS_TD_NB) {
				ncn = q.in[mem].cnt[tid];
				ns = q.in[mem].sum[tid];
			}
			uint64_t nold, nnew;
			const uint32_t no = compact_256(ns, ncn, n_sum, n_cnt, n_wpfx, &nnew, s_wv, s_ww);
			if (nnew) {
				const uint32_t nd = compact_256(d_sum[tid], d_cnt[tid], c_sum, c_cnt, c_wpfx, &nold, s_wv, s_ww);
				const uint64_t twoN = 2ull * (nold + nnew);
				s_T[tid] = (tid >= 1u && tid < GYS_TD_NB) ? td_threshold(c_td_bnd[tid], twoN) : (tid ? ~0ull : 0ull);
				o_sum[tid] = 0;
				o_cnt[tid] = 0;
				__syncthreads();
				if (tid < nd) { // old cluster: preceded by the old weight before it and the new weight with mean strictly below its mean
					const uint64_t S = (uint64_t)c_sum[tid], Cc = c_cnt[tid];
					uint32_t lo = 0, hi = no; // first item i with NOT (s_i / c_i < S / C)
					while (lo < hi) {
						const uint32_t mid = (lo + hi) >> 1;
						if (mul_lt((uint64_t)n_sum[mid], Cc, S, n_cnt[mid])) lo = mid + 1; else hi = mid;
					}
					const uint64_t mid2 = 2ull * (c_wpfx[tid] + n_wpfx[lo]) + Cc;
					const uint32_t cl = cluster_of_u64(s_T, mid2);
					atomicAdd(&o_sum[cl], (unsigned long long)S);
					atomicAdd(&o_cnt[cl], (unsigned long long)Cc);
				}
				if (tid < no) { // new item: preceded by the new weight before it and the old weight with mean <= its mean
					const uint64_t sI = (uint64_t)n_sum[tid], cI = n_cnt[tid];
					uint32_t lo = 0, hi = nd; // first old cluster j with NOT (S_j / C_j <= s / c)
					while (lo < hi) {
						const uint32_t mid = (lo + hi) >> 1;
						if (mul_le((uint64_t)c_sum[mid], cI, sI, c_cnt[mid])) lo = mid + 1; else hi = mid;
					}
					const uint64_t mid2 = 2ull * (n_wpfx[tid] + c_wpfx[lo]) + cI;
					const uint32_t cl = cluster_of_u64(s_T, mid2);
					atomicAdd(&o_sum[cl], (unsigned long long)sI);
					atomicAdd(&o_cnt[cl], (unsigned long long)cI);
				}
				__syncthreads();
				d_sum[tid] = (int64_t)o_sum[tid];
				d_cnt[tid] = o_cnt[tid];
				if (tid == 0) { // the member's own extremes
					long long vmn, vmx;
					if (q.kind == 0) {
						const int2 mm = p.td_minmax[mem];
						vmn = mm.x;
						vmx = mm.y;
					} else {
						vmn = q.in[mem].vmin;
						vmx = q.in[mem].vmax;
					}
					if (vmn < s_mm[0]) s_mm[0] = vmn;
					if (vmx > s_mm[1]) s_mm[1] = vmx;
				}
				__syncthreads();
			}
			if (!npend) continue;
			// ---------------- step B: the member's buffered values as unit points (value bins, see k_digest_bins)
			uint32_t wd[VPT];
			{
				const uint32_t *pend = p.td_pend + (size_t)mem * p.pcap;
#pragma unroll
				for (uint32_t k = 0; k < VPT; ++k) {
					const uint32_t i = tid + 256u * k;
					wd[k] = i < npend ? pend[i] : 0u;
				}
			}
			const uint32_t nd = compact_256(d_sum[tid], d_cnt[tid], c_sum, c_cnt, c_wpfx, &nold, s_wv, s_ww);
#pragma unroll
			for (uint32_t k = 0; k < GYS_MB_BPT; ++k) s_bin[tid + 256u * k] = 0;
			s_thr[tid] = 0xFFFFFFFFu;
			o_sum[tid] = 0;
			o_cnt[tid] = 0;
			if (tid == 0) s_nbig = 0;
			const uint64_t twoN = 2ull * (nold + (uint64_t)npend);
			s_T[tid] = (tid >= 1u && tid < GYS_TD_NB) ? td_threshold(c_td_bnd[tid], twoN) : (tid ? ~0ull : 0ull);
			__syncthreads();
			uint32_t thr = 0;
			if (tid < nd) { // integer mean threshold: mean <= v  <=>  ceil(S / C) <= v
				const uint64_t S = (uint64_t)c_sum[tid], Cc = c_cnt[tid];
				thr = (uint32_t)((S + Cc - 1ull) / Cc);
				s_thr[tid] = thr;
				atomicAdd(&s_bin[mb_bin(thr)], 1u << 16);
			}
			uint32_t pos[VPT];
			int32_t lmin = INT32_MAX, lmax = INT32_MIN;
#pragma unroll
			for (uint32_t k = 0; k < VPT; ++k) {
				const uint32_t i = tid + 256u * k;
				pos[k] = 0;
				if (i >= npend) continue;
				const uint32_t uv = wd[k] >> GYS_ROW_BITS;
				pos[k] = atomicAdd(&s_bin[mb_bin(uv)], 1u) & 0xFFFFu;
				if (uv >= GYS_MB_EXACT) s_big[atomicAdd(&s_nbig, 1u)] = (i << 20) | uv;
				lmin = min(lmin, (int32_t)uv);
				lmax = max(lmax, (int32_t)uv);
			}
#pragma unroll
			for (int d = 32; d >= 1; d >>= 1) {
				lmin = min(lmin, __shfl_xor(lmin, d, 64));
				lmax = max(lmax, __shfl_xor(lmax, d, 64));
			}
			if (lane == 0) {
				if (lmin != INT32_MAX) atomicMin(&s_mm[0], (long long)lmin);
				if (lmax != INT32_MIN) atomicMax(&s_mm[1], (long long)lmax);
			}
			__syncthreads();
			{ // one packed scan over the bins: {values in lower bins : 16 | clusters at or below the bin : 16}
				uint32_t bv[GYS_MB_BPT], own = 0;
				const uint4 lo4 = ((const uint4 *)s_bin)[2u * tid], hi4 = ((const uint4 *)s_bin)[2u * tid + 1u];
				bv[0] = lo4.x; bv[1] = lo4.y; bv[2] = lo4.z; bv[3] = lo4.w;
				bv[4] = hi4.x; bv[5] = hi4.y; bv[6] = hi4.z; bv[7] = hi4.w;
#pragma unroll
				for (uint32_t k = 0; k < GYS_MB_BPT; ++k) own += bv[k];
				uint32_t sc = own;
#pragma unroll
				for (int d = 1; d < 64; d <<= 1) {
					const uint32_t t = __shfl_up(sc, d, 64);
					if ((int)lane >= d) sc += t;
				}
				if (lane == 63u) s_ws[wave] = sc;
				__syncthreads();
				uint32_t run = sc - own;
#pragma unroll
				for (uint32_t k = 0; k < 3u; ++k)
					if (k < wave) run += s_ws[k];
#pragma unroll
				for (uint32_t k = 0; k < GYS_MB_BPT; ++k) {
					const uint32_t raw = bv[k];
					bv[k] = (run & 0xFFFFu) | (((run >> 16) + (raw >> 16)) << 16);
					run += raw;
				}
				((uint4 *)s_bin)[2u * tid] = make_uint4(bv[0], bv[1], bv[2], bv[3]);
				((uint4 *)s_bin)[2u * tid + 1u] = make_uint4(bv[4], bv[5], bv[6], bv[7]);
			}
			__syncthreads();
			const uint32_t nbig = s_nbig;
			if (tid < nd) {
				uint32_t nb = s_bin[mb_bin(thr)] & 0xFFFFu;
				if (thr >= GYS_MB_EXACT) {
					const uint32_t sh = (31u - (uint32_t)__clz((int)thr)) - 6u;
					for (uint32_t j = 0; j < nbig; ++j) {
						const uint32_t u = s_big[j] & 0xFFFFFu;
						nb += ((u >> sh) == (thr >> sh) && u < thr) ? 1u : 0u;
					}
				}
				const uint64_t mid2 = 2ull * (c_wpfx[tid] + (uint64_t)nb) + c_cnt[tid];
				const uint32_t cl = cluster_of_u64(s_T, mid2);
				atomicAdd(&o_sum[cl], (unsigned long long)c_sum[tid]);
				atomicAdd(&o_cnt[cl], (unsigned long long)c_cnt[tid]);
			}
#pragma unroll
			for (uint32_t k = 0; k < VPT; ++k) {
				const uint32_t i = tid + 256u * k;
				const uint32_t uv = wd[k] >> GYS_ROW_BITS;
				if (i >= npend || uv >= GYS_MB_EXACT) continue;
				const uint32_t bw = s_bin[uv];
				const uint64_t mid2 = 2ull * ((uint64_t)((bw & 0xFFFFu) + pos[k]) + c_wpfx[bw >> 16]) + 1ull;
				const uint32_t cl = cluster_of_u64(s_T, mid2);
				atomicAdd(&o_sum[cl], (unsigned long long)uv);
				atomicAdd(&o_cnt[cl], 1ull);
			}
			for (uint32_t j = tid; j < nbig; j += 256u) {
				const uint32_t me = s_big[j], uv = me & 0xFFFFFu, i = me >> 20;
				const uint32_t sh = (31u - (uint32_t)__clz((int)uv)) - 6u;
				uint32_t r = s_bin[mb_bin(uv)] & 0xFFFFu;
				for (uint32_t jj = 0; jj < nbig; ++jj) {
					const uint32_t e = s_big[jj], u = e & 0xFFFFFu;
					r += ((u >> sh) == (uv >> sh) && (u < uv || (u == uv && (e >> 20) < i))) ? 1u : 0u;
				}
				uint32_t gap = 0;
#pragma unroll
				for (uint32_t step = GYS_NBP / 2; step >= 1u; step >>= 1)
					if (s_thr[gap + step - 1u] <= uv) gap += step;
				const uint64_t mid2 = 2ull * ((uint64_t)r + c_wpfx[gap]) + 1ull;
				const uint32_t cl = cluster_of_u64(s_T, mid2);
				atomicAdd(&o_sum[cl], (unsigned long long)uv);
				atomicAdd(&o_cnt[cl], 1ull);
			}
			__syncthreads();
			d_sum[tid] = (int64_t)o_sum[tid];
			d_cnt[tid] = o_cnt[tid];
			__syncthreads();
		}
		if (tid < GYS_TD_NB) {
			q.out[g].sum[tid] = d_sum[tid];
			q.out[g].cnt[tid] = d_cnt[tid];
		}
		if (tid == 0) {
			q.out[g].vmin = s_mm[0];
			q.out[g].vmax = s_mm[1];
		}
		__syncthreads();
	}
}

} // namespace gys
